export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { label=$1; shift
env "$@" > gpurun_out/x.json 2> gpurun_out/x.err || tail -3 gpurun_out/x.err
python - "$label" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1], 'value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()})
PY
}
B="timeout 600 python bench.py --workload e2e --steps 60 --warmup 2"
( run "default (3 push streams, fold on the decoder's stream)" X=1 $B
run "two-thread consumer (fold on its own stream: 2 in rotation)" X=1 $B --two-thread-consumer --parity none
run "two-thread consumer, 3 pinned" SURGE_INGEST_PUSH_STREAMS=3 $B --two-thread-consumer --parity none ) 2>&1 | tee gpurun_out/r06_e2e_push_streams_auto.txt
bash scripts/experiments/run_r6_suite.sh
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; tail -4 gpurun_out/r06_bench_n1.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value %.4e'%d['value'], 'frac', round(r['frac'],4), 'one_shot', round(r.get('frac_one_shot',0),4), 'traffic', r.get('traffic'), 'kernel', r.get('kernel'), 'kernel_ms', r['kernel_ms'], 'probe', r['stream_read_probe_GBps'])
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
print('cpu_baseline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['cpu_baseline'].items() if not isinstance(v,(dict,list,str))})
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
