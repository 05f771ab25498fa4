mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_ingest_gpu.py tests/test_store.py "tests/test_bench_rehearsal.py::test_bench_workload_e2e_goes_from_topic_bytes_to_states_and_checks_them_against_the_source_events" "tests/test_bench_rehearsal.py::test_bench_workload_e2e_shards_the_ingest_by_partition_over_the_ranks" -x -q -m gpu > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4i/pytest.log
for extra in "" "--no-capacity-hint"; do
timeout 900 python bench.py --workload e2e $extra > gpurun_out/r4i/e2e$extra.json 2> gpurun_out/r4i/e2e$extra.err; echo "e2e rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4i/e2e$extra.json')); c=d['config']; print('$extra', d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['finish_and_fold_ms_per_fetch'], c['push_async_host_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])"
done
