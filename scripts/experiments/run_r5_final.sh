#!/bin/bash
# Round 5, closing run on the final sources: the GPU suite, smoke(), the default line (what the driver runs), kernel traces of
# the bytes -> states path pipelined and one push at a time.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5final; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; lap "pytest rc=$?"
tail -n 22 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; lap "smoke rc=$?"
tail -n 2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; lap "bench rc=$?"
python - <<'P'
import json
O="gpurun_out/r5final"
try:
    d=json.loads([l for l in open(O+"/bench_n1.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("default:", d["config"]["algo"], r["kernel"], "%.4g"%d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "probe", r.get("stream_read_probe_GBps"))
    print("tile_major:", d["tile_major"]["frac"], d["tile_major"]["kernel_ms_min_median_max"])
    print("secondary:", d["secondary"]["roofline"]["frac"], d["secondary"].get("tile_major",{}).get("frac"))
    c4=d["c4_shard"]; print("c4_shard:", c4["roofline"]["frac"], c4["roofline"]["traffic"], c4["roofline"]["kernel_ms_min_median_max"], c4["cpu_baseline"])
    e=d.get("e2e",{}); c=e.get("config",{})
    print("e2e:", e.get("value"), e.get("skipped"), json.dumps(e.get("layouts_events_per_s")))
    print("e2e cfg:", {k:c.get(k) for k in ("fetch_ms","host_framing_ms_per_fetch","finish_and_fold_ms_per_fetch","push_async_host_ms_per_fetch","framing_threads","events_timed","generate_s","parity_s")})
    print("c5", d["c5"].get("value"), d["c5"].get("ms_per_step"), "v2", d["v2"].get("roofline",{}).get("frac"))
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_matches_cpu_full_log"])
except Exception as e: print("default failed", e)
P
PROF_PASSES=trace PROF_PASS_TIMEOUT=150 timeout 200 python scripts/prof_ingest.py r05_e2e_k512 --workload e2e --steps 12 --warmup 2 --txn-flush-events 512 > $O/prof_d4.log 2>&1; lap "prof depth4 rc=$?"
head -n 22 gpurun_out/prof_r05_e2e_k512/*_summary.txt
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=150 timeout 200 python scripts/prof_ingest.py r05_e2e_k512_depth1 --workload e2e --steps 12 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_d1.log 2>&1; lap "prof depth1 rc=$?"
head -n 26 gpurun_out/prof_r05_e2e_k512_depth1/*_summary.txt
