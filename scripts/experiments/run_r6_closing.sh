export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; tail -4 gpurun_out/r06_bench_n1.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value %.4e'%d['value'], 'frac', round(r['frac'],4), 'one_shot', round(r.get('frac_one_shot',0),4), 'traffic', r.get('traffic'), 'kernel', r.get('kernel'))
for k in ('secondary','c4_shard','tile_major','v2','c5'):
    o=d.get(k)
    if isinstance(o,dict): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in o.items() if not isinstance(vv,(dict,list,str))}, (o.get('roofline') or {}).get('frac'))
e=d.get('e2e',{})
print('e2e value %.4e'%e.get('value',0), 'layouts', e.get('layouts_events_per_s'))
print('bound_log', {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.get('bound_log',{}).items() if not isinstance(v,(dict,list))})
print('by_copy', e.get('framing_by_copy_12_threads'))
print('8thr', e.get('in_place_8_threads'))
print('mixed', {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.get('mixed_topic',{}).items() if k!='workload'})
c=e.get('config',{})
print('host cpu', c.get('host_cpu_ms_per_1e6_records'), c.get('host_cpu_ms_per_1e6_records_without_the_receive_copy'), c.get('host_cpu_ms_per_1e6_records_by_thread'), c.get('consumer_cpu_ms_per_fetch'), c.get('fetch_ms'))
print('cpu_baseline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['cpu_baseline'].items() if not isinstance(v,(dict,list,str))})
PY
bash scripts/experiments/run_r6_suite.sh
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
PROF_PASSES=trace,sq1 PROF_PASS_TIMEOUT=200 timeout 600 python scripts/prof_ingest.py r06_e2e_closing --workload e2e --steps 12 --warmup 2 --parity none 2>&1 | tail -30
rm -rf gpurun_out/prof_r06_e2e_closing/trace gpurun_out/prof_r06_e2e_closing/sq1
run() { tag=$1; shift; timeout 600 "$@" > gpurun_out/r06_host_$tag.json 2> gpurun_out/r06_host_$tag.err; tail -2 gpurun_out/r06_host_$tag.err | cut -c1-300
python - $tag <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r06_host_{sys.argv[1]}.json').read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1], 'no line', e); raise SystemExit
c=d['config']
print(sys.argv[1], 'aggregate %.3e records/s'%d['value'], 'ranks', c['ranks'], 'threads/rank', c['framing_threads_per_rank'], 'slabs MB', c['slab_bytes_all_ranks']/1e6, 'pinned', c['page_locked_slabs'])
for r in c['per_rank'][:8]: print('   ', {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
PY
}
run n1_inplace2 python bench.py --workload e2e --host-only --steps 12 --warmup 2 --framing-threads 2
SURGE_BENCH_REHEARSAL=1 run n8_inplace python bench.py --workload e2e --gpus 8 --host-only --steps 12 --warmup 2 --framing-threads 2
