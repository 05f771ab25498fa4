export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu --timeout 200 2>&1 | grep -E "passed|failed" | tail -1
for rep in 1 2; do
for mode in block poll spin; do
for t in 2 3; do
SURGE_INGEST_WAIT=$mode timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads $t --parity none > gpurun_out/x.json 2> gpurun_out/x.err
python - $mode $t <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
cc=c['consumer_cpu_ms_per_fetch']
print('waits', sys.argv[1], 'threads', sys.argv[2], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv', round(c['receive_copy_cpu_ms_per_1e6_records'],2), 'framing', round(c['framing_cpu_ms_per_1e6_records'],2), 'consumer cpu push/finish/fold', [round(v,3) for v in list(cc.values())[:3]], 'by thread', c['host_cpu_ms_per_1e6_records_by_thread'], 'p50', round(c['fetch_ms']['p50'],2))
PY
done
done
done 2>&1 | tee gpurun_out/r06_e2e_wait_modes.txt
SURGE_INGEST_WAIT=poll timeout 400 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 200 2>&1 | grep -E "passed|failed" | tail -1
