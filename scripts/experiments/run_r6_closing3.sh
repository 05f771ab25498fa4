export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; tail -4 gpurun_out/r06_bench_n1.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value %.4e'%d['value'], 'frac', round(r['frac'],4), 'one_shot', round(r.get('frac_one_shot',0),4), 'traffic', r.get('traffic'), 'kernel_ms', r['kernel_ms'], 'probe', r['stream_read_probe_GBps'])
for k in ('secondary','c4_shard','tile_major','v2','c5'):
    o=d.get(k)
    if isinstance(o,dict): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in o.items() if not isinstance(vv,(dict,list,str))}, (o.get('roofline') or {}).get('frac'))
print('c2 tiled', d['secondary']['tile_major']['frac'], 'c5 lat', d['c5']['config']['batch_latency_ms'])
e=d.get('e2e',{})
print('e2e value %.4e'%e.get('value',0), 'layouts', e.get('layouts_events_per_s'))
print('bound_log', {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.get('bound_log',{}).items() if not isinstance(v,(dict,list))})
print('by_copy', e.get('framing_by_copy_12_threads'))
print('8thr', e.get('in_place_8_threads'))
print('mixed', {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.get('mixed_topic',{}).items() if k!='workload'})
c=e.get('config',{})
print('host cpu', c.get('host_cpu_ms_per_1e6_records'), c.get('host_cpu_ms_per_1e6_records_without_the_receive_copy'), c.get('framing_cpu_ms_per_1e6_records'), c.get('receive_copy_cpu_ms_per_1e6_records'), c.get('host_cpu_ms_per_1e6_records_by_thread'), c.get('consumer_cpu_ms_per_fetch'), c.get('fetch_ms'))
PY
timeout 300 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 200 2>&1 | grep -E "passed|failed" | tail -1
