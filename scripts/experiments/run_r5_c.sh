#!/bin/bash
# Round 5, third run: push slots sized up front (no allocation in the pipeline): the three layouts again, per-fetch trace;
# the device stage's kernels one push at a time (depth 1) for their un-overlapped durations.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5c; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
for K in 512 64 0; do
  SURGE_BENCH_TRACE=1 timeout 300 python bench.py --workload e2e --steps 28 --warmup 2 --txn-flush-events $K > $O/e2e_k$K.json 2> $O/e2e_k$K.err; lap "e2e K=$K rc=$?"
done
python - <<'P'
import json
O="gpurun_out/r5c"
for n in ("k512","k64","k0"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "parity", d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"], "fetch_ms", c["fetch_ms"], "framing ms", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2), "disc %.3g known %.3g"%(c["events_per_s_while_discovering_keys"] or 0, c["events_per_s_all_keys_known"] or 0))
    except Exception as e: print(n, "failed", e)
P
grep "\[bench\] fetch" $O/e2e_k512.err | head -n 34
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05_e2e_k512_depth1 --workload e2e --steps 10 --warmup 2 --txn-flush-events 512 > $O/prof_k512_d1.log 2>&1; lap "prof k512 depth1 rc=$?"
head -n 24 gpurun_out/prof_r05_e2e_k512_depth1/*_summary.txt
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05_e2e_k64_depth1 --workload e2e --steps 10 --warmup 2 --txn-flush-events 64 > $O/prof_k64_d1.log 2>&1; lap "prof k64 depth1 rc=$?"
head -n 24 gpurun_out/prof_r05_e2e_k64_depth1/*_summary.txt
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_ingest.log 2>&1; lap "pytest ingest rc=$?"
tail -n 3 $O/pytest_ingest.log
