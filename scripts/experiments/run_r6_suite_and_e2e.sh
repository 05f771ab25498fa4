export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x 2>&1 | tail -4 | tee gpurun_out/r06_gpu_suite.txt
for rep in 1 2; do
timeout 600 python bench.py --workload e2e --steps 100 --warmup 2 > gpurun_out/x.json 2> gpurun_out/x.err || tail -3 gpurun_out/x.err
python - <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print('default e2e: value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], 'threads', c['framing_threads'], 'consumer', c['consumer'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'w/o recv', round(c['host_cpu_ms_per_1e6_records_without_the_receive_copy'],2), 'by thread', c['host_cpu_ms_per_1e6_records_by_thread'], 'consumer cpu', c['consumer_cpu_ms_per_fetch'], 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()})
PY
done 2>&1 | tee gpurun_out/r06_e2e_new_defaults.txt
