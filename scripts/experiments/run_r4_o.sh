mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu > gpurun_out/r4o/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4o/pytest.log | cut -c1-300
PROF_PASSES=trace PROF_PASS_TIMEOUT=600 timeout 900 python scripts/prof_ingest.py r04_e2e_serial --workload e2e --events-cap 2 --serial-framing > gpurun_out/r4o/prof.log 2>&1; echo "prof rc=$?"
head -10 gpurun_out/prof_r04_e2e_serial/r04_e2e_serial_summary.txt
timeout 900 python bench.py --workload e2e > gpurun_out/r4o/e2e.json 2> gpurun_out/r4o/e2e.err; echo "e2e rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4o/e2e.json')); c=d['config']; print(d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['finish_and_fold_ms_per_fetch'], c['push_async_host_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])"
