#!/bin/bash
# Round 5, last run (3 GPU-minutes left): the ingest group sizes all six page-locked slabs at its first feed — the ingest / store
# GPU tests, the small-flush layout three times (two of a dozen earlier runs lost a 24 ms fetch), the default layout once.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5l; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 100 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_ingest.log 2>&1; lap "pytest ingest rc=$?"
tail -n 2 $O/pytest_ingest.log
for i in 1 2 3; do
  SURGE_BENCH_TRACE=1 timeout 60 python bench.py --workload e2e --warmup 2 --parity none --steps 10 --txn-flush-events 64 > $O/e2e_k64_$i.json 2> $O/e2e_k64_$i.err; lap "e2e k64 $i rc=$?"
done
timeout 60 python bench.py --workload e2e --warmup 2 --parity none --steps 28 --txn-flush-events 512 > $O/e2e_k512.json 2> $O/e2e_k512.err; lap "e2e k512 rc=$?"
python - <<'P'
import json
O="gpurun_out/r5l"
for n in ("k64_1","k64_2","k64_3","k512"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "fetch_ms", {k: round(v,2) for k,v in c["fetch_ms"].items()}, "framing", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2))
    except Exception as e: print(n, "failed", e)
P
grep "framing ms per fetch\|ms between" $O/e2e_k64_1.err
