mkdir -p gpurun_out/r4d
timeout 1200 python -m pytest tests/test_ingest_gpu.py tests/test_store.py "tests/test_bench_rehearsal.py::test_bench_workload_e2e_goes_from_topic_bytes_to_states_and_checks_them_against_the_source_events" "tests/test_bench_rehearsal.py::test_bench_workload_e2e_shards_the_ingest_by_partition_over_the_ranks" -x -q -m gpu > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r4d/pytest.log
timeout 900 python bench.py --workload e2e > gpurun_out/r4d/e2e.json 2> gpurun_out/r4d/e2e.err; echo "e2e rc=$?"; tail -5 gpurun_out/r4d/e2e.err
python -c "
import json; d=json.load(open('gpurun_out/r4d/e2e.json')); c=d['config']; print(d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['finish_and_fold_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], c['generate_s'], c['parity_s'], c['keys_interned'], c['wire_bytes_per_record'], d['cpu_baseline'])"
