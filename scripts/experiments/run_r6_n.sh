export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "short_rows or random_log_shapes or empty or ragged or zipf or auto" 2>&1 | tail -6
timeout 600 python bench.py --workload e2e --steps 12 --warmup 2 --bound-log > gpurun_out/r06_e2e_bound_log.json 2> gpurun_out/r06_e2e_bound_log.err; tail -3 gpurun_out/r06_e2e_bound_log.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e2e_bound_log.json').read().strip().splitlines()[-1])
print('value', d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])
print(json.dumps(d['config']['bound_log']))
PY
