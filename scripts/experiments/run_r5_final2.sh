#!/bin/bash
# Round 5, after the closing run (same sources): every counter pass of the headline fold and of the C4 shard's on the final
# sources (the concrete walk), the SQ passes of the bytes -> states kernels one push at a time, and a second default line.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5final2; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
PROF_PASS_TIMEOUT=120 timeout 400 python scripts/prof_traffic.py r05_c3_10Magg_sorted --parity none > $O/prof_c3_sorted.log 2>&1; lap "prof c3 sorted rc=$?"
grep -E "fold_sorted_pf|traffic per launch" $O/prof_c3_sorted.log | cut -c1-200
PROF_PASS_TIMEOUT=100 timeout 300 python scripts/prof_traffic.py r05_c4shard_auto --workload c4-shard > $O/prof_c4shard.log 2>&1; lap "prof c4shard rc=$?"
grep -E "fold_chunked|traffic per launch" $O/prof_c4shard.log | cut -c1-200
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace,sq1,sq2 PROF_PASS_TIMEOUT=120 timeout 300 python scripts/prof_ingest.py r05_e2e_k512_depth1 --workload e2e --steps 10 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_d1.log 2>&1; lap "prof depth1 rc=$?"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; lap "bench rc=$?"
python - <<'P'
import json
O="gpurun_out/r5final2"
try:
    d=json.loads([l for l in open(O+"/bench_n1.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("default:", d["config"]["algo"], r["kernel"], "%.4g"%d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "probe", r.get("stream_read_probe_GBps"))
    print("tile_major:", d["tile_major"]["frac"]); print("secondary:", d["secondary"]["roofline"]["frac"], d["secondary"].get("tile_major",{}).get("frac"))
    print("c4_shard:", d["c4_shard"]["roofline"]["frac"]); e=d.get("e2e",{}); print("e2e:", e.get("value"), json.dumps(e.get("layouts_events_per_s"))); print("c5", d["c5"].get("value"), "v2", d["v2"].get("roofline",{}).get("frac"))
except Exception as e: print("default failed", e)
P
