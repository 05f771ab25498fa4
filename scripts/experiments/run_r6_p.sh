export TMPDIR=/tmp
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { tag=$1; shift; timeout 900 "$@" > gpurun_out/r06_host_$tag.json 2> gpurun_out/r06_host_$tag.err; tail -2 gpurun_out/r06_host_$tag.err | cut -c1-300
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
run n1_inplace3 python bench.py --workload e2e --host-only --steps 12 --warmup 2 --framing-threads 3
run n1_copy12 python bench.py --workload e2e --host-only --steps 12 --warmup 2 --framing-threads 12 --framing-by-copy
SURGE_BENCH_REHEARSAL=1 run n8_inplace python bench.py --workload e2e --gpus 8 --host-only --steps 12 --warmup 2 --framing-threads 3
SURGE_BENCH_REHEARSAL=1 run n8_copy python bench.py --workload e2e --gpus 8 --host-only --steps 12 --warmup 2 --framing-threads 3 --framing-by-copy
