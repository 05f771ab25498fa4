mkdir -p gpurun_out/r4h
timeout 900 python bench.py --workload e2e > gpurun_out/r4h/e2e.json 2> gpurun_out/r4h/e2e.err; echo "e2e rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4h/e2e.json')); c=d['config']; print(d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['finish_and_fold_ms_per_fetch'], c['push_async_host_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])"
