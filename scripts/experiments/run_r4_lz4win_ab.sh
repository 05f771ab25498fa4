#!/bin/bash
# the two window loops of lz4_exec_kernel on one box, back to back
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4lz4win; mkdir -p $O
SURGE_INGEST_LZ4_WINDOWS=1 timeout 120 python bench.py --workload e2e > $O/e2e_one_window.json 2> $O/e2e_one_window.err; echo "one-window rc=$?" | tee -a $O/rc.log
timeout 120 python bench.py --workload e2e > $O/e2e_batched.json 2> $O/e2e_batched.err; echo "batched rc=$?" | tee -a $O/rc.log
python - <<'P'
import json
for n in ("e2e_one_window","e2e_batched"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r4lz4win/{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g"%d["value"], d["ms_per_step"], "steady %.4g"%c["events_per_s_all_keys_known"], d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"])
    except Exception as e: print(n, "failed", e)
P
