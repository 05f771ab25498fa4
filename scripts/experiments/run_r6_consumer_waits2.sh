export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in finish:2 finish:3 none:3 finish:4; do
w=${cfg%%:*}; t=${cfg##*:}
SURGE_BENCH_TRACE=1 timeout 600 python bench.py --workload e2e --steps 40 --warmup 2 --framing-threads $t --consumer-waits $w --parity none > gpurun_out/x.json 2> gpurun_out/x_$cfg.err
python - $cfg <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
cc=c['consumer_cpu_ms_per_fetch']
print('cfg', sys.argv[1], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'recv wall', round(c['receive_copy_ms_per_fetch'],2), 'framing wall', round(c['host_framing_ms_per_fetch'],2), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()}, 'finish+fold wall', round(c['finish_and_fold_ms_per_fetch'],2), 'push wall', round(c['push_async_host_ms_per_fetch'],2))
PY
grep "bench\] fetch" gpurun_out/x_$cfg.err | sed -n 10,30p
grep "ms between\|framing ms\|push_async host" gpurun_out/x_$cfg.err | cut -c1-400
done 2>&1 | tee gpurun_out/r06_e2e_consumer_waits_trace.txt
