export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 200 2>&1 | grep -E "passed|failed" | tail -1
for t in 3 2 3 2; do
for flush in 512 64; do
timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads $t --txn-flush-events $flush --parity none > gpurun_out/x.json 2> gpurun_out/x.err
python - $t $flush <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print('threads', sys.argv[1], 'flush', sys.argv[2], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv', round(c['receive_copy_cpu_ms_per_1e6_records'],2), 'framing', round(c['framing_cpu_ms_per_1e6_records'],2), 'framing wall/fetch', round(c['host_framing_ms_per_fetch'],2), 'p50', round(c['fetch_ms']['p50'],2))
PY
done
done 2>&1 | tee gpurun_out/r06_e2e_header_prewalk.txt
