export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
for mode in block poll:25 poll:10 poll:50; do
m=${mode%%:*}; us=${mode##*:}
SURGE_INGEST_WAIT=$m SURGE_INGEST_POLL_US=$us timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads 2 --parity none > gpurun_out/x.json 2> gpurun_out/x.err
python - $mode <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
cc=c['consumer_cpu_ms_per_fetch']
print('waits', sys.argv[1], 'threads 2 value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv', round(c['receive_copy_cpu_ms_per_1e6_records'],2), 'framing', round(c['framing_cpu_ms_per_1e6_records'],2), 'consumer cpu push/finish/fold', [round(v,3) for v in list(cc.values())[:3]], 'main', c['host_cpu_ms_per_1e6_records_by_thread'].get('MainThread'), 'p50', round(c['fetch_ms']['p50'],2))
PY
done
done 2>&1 | tee gpurun_out/r06_e2e_wait_modes2.txt
