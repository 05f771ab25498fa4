#!/bin/bash
# two-thread consumer + event-ordered hand-over: tests, then the e2e line both ways on the same box
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4pusher; mkdir -p $O
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" | tee -a $O/rc.log
timeout 600 python -m pytest tests/test_bench_rehearsal.py -m gpu -x -q -k e2e > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?" | tee -a $O/rc.log
tail -3 $O/pytest_a.log $O/pytest_b.log
SURGE_BENCH_TRACE=1 timeout 400 python bench.py --workload e2e > $O/e2e_new.json 2> $O/e2e_new.err; echo "e2e_new rc=$?" | tee -a $O/rc.log
timeout 400 python bench.py --workload e2e --one-thread-consumer > $O/e2e_old.json 2> $O/e2e_old.err; echo "e2e_old rc=$?" | tee -a $O/rc.log
SURGE_BENCH_DEPTH=5 timeout 400 python bench.py --workload e2e > $O/e2e_new_d5.json 2> $O/e2e_new_d5.err; echo "e2e_new_d5 rc=$?" | tee -a $O/rc.log
python - <<'P'
import json
for n in ["e2e_new","e2e_old","e2e_new_d5"]:
    try:
        d=json.loads([l for l in open(f"gpurun_out/r4pusher/{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g"%d["value"], "ms/step %.3f"%d["ms_per_step"], "push %.3f finish %.3f framing %.3f"%(c["push_async_host_ms_per_fetch"],c["finish_and_fold_ms_per_fetch"],c["host_framing_ms_per_fetch"]), "steady %.4g"%(c["events_per_s_all_keys_known"] or 0), d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"])
    except Exception as e: print(n, "failed", e)
P
tail -5 $O/e2e_new.err
