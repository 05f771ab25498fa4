export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -m gpu --timeout 200 -k "point_reads or jni" 2>&1 | tail -2
for arm in stream memcpy stream memcpy; do
if [ $arm = memcpy ]; then export SURGE_INGEST_RECEIVE_COPY=memcpy; else unset SURGE_INGEST_RECEIVE_COPY; fi
for t in 3 2; do
timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads $t --parity none > gpurun_out/x.json 2> gpurun_out/x.err
python - $arm $t <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1], 'threads', sys.argv[2], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv', round(c['receive_copy_cpu_ms_per_1e6_records'],2), 'framing', round(c['framing_cpu_ms_per_1e6_records'],2), 'recv wall/fetch', round(c['receive_copy_ms_per_fetch'],2), 'framing wall/fetch', round(c['host_framing_ms_per_fetch'],2))
PY
done
done 2>&1 | tee gpurun_out/r06_e2e_receive_copy_ab.txt
