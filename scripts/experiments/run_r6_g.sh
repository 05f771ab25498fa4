export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --workload e2e --steps 12 --warmup 2 --bound-log > gpurun_out/r06_e2e_bound_log.json 2> gpurun_out/r06_e2e_bound_log.err; tail -3 gpurun_out/r06_e2e_bound_log.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e2e_bound_log.json').read().strip().splitlines()[-1])
print('value', d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])
print(json.dumps(d['config']['bound_log']))
print({k:d['config'][k] for k in ('host_cpu_ms_per_1e6_records','framing_threads','host_framing_ms_per_fetch','finish_and_fold_ms_per_fetch','push_async_host_ms_per_fetch','fetch_ms')})
PY
timeout 900 python bench.py --workload e2e --steps 12 --warmup 2 > gpurun_out/r06_e2e_k3.json 2> gpurun_out/r06_e2e_k3.err; tail -3 gpurun_out/r06_e2e_k3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e2e_k3.json').read().strip().splitlines()[-1])
print('value', d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])
print({k:d['config'][k] for k in ('host_cpu_ms_per_1e6_records','framing_threads','host_framing_ms_per_fetch','finish_and_fold_ms_per_fetch','push_async_host_ms_per_fetch','fetch_ms')})
PY
