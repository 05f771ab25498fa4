mkdir -p gpurun_out/r4depth
for depth in 3 4 5 2; do
SURGE_BENCH_DEPTH=$depth timeout 600 python bench.py --workload e2e > gpurun_out/r4depth/e2e_d$depth.json 2>/dev/null; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4depth/e2e_d$depth.json')); c=d['config']; print('depth $depth', d['value'], c['fetch_ms'], c['finish_and_fold_ms_per_fetch'], c['push_async_host_ms_per_fetch'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])"
done
