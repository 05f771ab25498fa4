#!/bin/bash
# lz4_exec_kernel: eight windows' map reads in flight together, dependency rounds by ballot bit instead of a cross-lane read
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4lz4win; mkdir -p $O
timeout 300 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.log
grep -n "passed\|failed" $O/pytest.log | tail -n 2
timeout 300 python bench.py --workload e2e > $O/e2e.json 2> $O/e2e.err; echo "e2e rc=$?" | tee -a $O/rc.log
python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4lz4win/e2e.json") if l.startswith("{")][-1]); c=d["config"]
    print("e2e: %.4g"%d["value"], d["ms_per_step"], c["fetch_ms"], "steady %.4g"%c["events_per_s_all_keys_known"], d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"])
except Exception as e: print("e2e failed", e)
P
