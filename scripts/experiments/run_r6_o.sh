export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py --workload e2e --bound-log --aggregates 250000 --events-cap 4096 --steps 400 --warmup 2 > gpurun_out/r06_e2e_bound_log_long_rows.json 2> gpurun_out/r06_e2e_bound_log_long_rows.err; tail -3 gpurun_out/r06_e2e_bound_log_long_rows.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e2e_bound_log_long_rows.json').read().strip().splitlines()[-1])
print('value', d['value'], 'steps', d['steps'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], 'events', d['config']['events_timed'], 'gen_s', d['config']['generate_s'])
print(json.dumps(d['config']['bound_log']))
PY
