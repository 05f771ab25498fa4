export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
for w in both finish none; do
timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads 2 --consumer-waits $w > gpurun_out/x.json 2> gpurun_out/x.err || tail -5 gpurun_out/x.err
python - $w <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
cc=c['consumer_cpu_ms_per_fetch']
print('consumer waits', sys.argv[1], 'threads 2 value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv', round(c['receive_copy_cpu_ms_per_1e6_records'],2), 'framing', round(c['framing_cpu_ms_per_1e6_records'],2), 'consumer cpu push/finish/fold', [round(v,3) for v in list(cc.values())[:3]], 'main', c['host_cpu_ms_per_1e6_records_by_thread'].get('MainThread'), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()}, 'finish+fold wall', round(c['finish_and_fold_ms_per_fetch'],2), 'push wall', round(c['push_async_host_ms_per_fetch'],2))
PY
done
done 2>&1 | tee gpurun_out/r06_e2e_consumer_waits.txt
