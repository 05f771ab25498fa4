mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_ingest_gpu.py "tests/test_bench_rehearsal.py::test_bench_workload_e2e_goes_from_topic_bytes_to_states_and_checks_them_against_the_source_events" -x -q -m gpu > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4e/pytest.log
for t in 4 8 16; do
timeout 900 python bench.py --workload e2e --framing-threads $t > gpurun_out/r4e/e2e_t$t.json 2> gpurun_out/r4e/e2e_t$t.err; echo "e2e rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4e/e2e_t$t.json')); c=d['config']; print($t, d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['finish_and_fold_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'])"
done
