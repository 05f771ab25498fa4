#!/bin/bash
# Round 5, first run: topic bytes -> states on reference-shaped topics (independent writer, one transaction per flush), the three
# layouts beside each other, kernel traces of the small-flush and the default layout, the ingest / store GPU tests.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5a; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
for K in 512 64 0; do
  timeout 300 python bench.py --workload e2e --steps 12 --warmup 2 --txn-flush-events $K > $O/e2e_k$K.json 2> $O/e2e_k$K.err; lap "e2e K=$K rc=$?"
done
timeout 300 python bench.py --workload e2e --steps 12 --warmup 2 --writer product > $O/e2e_product.json 2> $O/e2e_product.err; lap "e2e product rc=$?"
python - <<'P'
import json
O="gpurun_out/r5a"
for n in ("k512","k64","k0","product"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "parity", d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"], "topic", c.get("topic"), "framing ms", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2), "gen_s", round(c["generate_s"],1))
    except Exception as e: print(n, "failed", e)
P
PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05_e2e_k64 --workload e2e --steps 6 --warmup 2 --txn-flush-events 64 > $O/prof_k64.log 2>&1; lap "prof k64 rc=$?"
PROF_PASSES=trace PROF_PASS_TIMEOUT=200 timeout 300 python scripts/prof_ingest.py r05_e2e_k512 --workload e2e --steps 6 --warmup 2 --txn-flush-events 512 > $O/prof_k512.log 2>&1; lap "prof k512 rc=$?"
cat gpurun_out/prof_r05_e2e_k64/*_summary.txt 2>/dev/null | head -40
cat gpurun_out/prof_r05_e2e_k512/*_summary.txt 2>/dev/null | head -40
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_ingest.log 2>&1; lap "pytest ingest rc=$?"
tail -n 3 $O/pytest_ingest.log
