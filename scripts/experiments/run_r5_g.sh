#!/bin/bash
# Round 5, seventh run (after a container restart lost runs c-f's outputs): the state of HEAD — the GPU suite, the default line
# (e2e with its layouts and c4_shard ride on it), a kernel trace of the e2e path at its default depth and one push at a time.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5g; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; lap "pytest rc=$?"
tail -n 4 $O/pytest.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; lap "bench rc=$?"
python - <<'P'
import json
O="gpurun_out/r5g"
try:
    d=json.loads([l for l in open(O+"/bench_n1.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("default:", d["config"]["algo"], r["kernel"], "%.4g"%d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "probe", r.get("stream_read_probe_GBps"))
    print("tile_major:", json.dumps(d.get("tile_major"))[:600])
    print("secondary:", json.dumps(d.get("secondary",{}).get("roofline"))[:600])
    print("c4_shard:", json.dumps(d.get("c4_shard"))[:1200])
    e=d.get("e2e",{}); c=e.get("config",{})
    print("e2e:", e.get("value"), e.get("skipped"), json.dumps(e.get("layouts_events_per_s")))
    print("e2e cfg:", {k:c.get(k) for k in ("fetch_ms","host_framing_ms_per_fetch","finish_and_fold_ms_per_fetch","push_async_host_ms_per_fetch","events_per_s_all_keys_known","events_per_s_while_discovering_keys","events_timed","generate_s")})
    print("c5", d["c5"].get("value"), d["c5"].get("ms_per_step"), "v2", d["v2"].get("roofline",{}).get("frac"))
except Exception as e: print("default failed", e)
P
PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05g_e2e_k512 --workload e2e --steps 12 --warmup 2 --txn-flush-events 512 > $O/prof_k512.log 2>&1; lap "prof k512 rc=$?"
head -n 30 gpurun_out/prof_r05g_e2e_k512/*_summary.txt
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05g_e2e_k512_depth1 --workload e2e --steps 12 --warmup 2 --txn-flush-events 512 > $O/prof_k512_d1.log 2>&1; lap "prof k512 depth1 rc=$?"
head -n 30 gpurun_out/prof_r05g_e2e_k512_depth1/*_summary.txt
