export TMPDIR=/tmp
mkdir -p gpurun_out
run() { label=$1; shift
env "$@" > gpurun_out/x.json 2> gpurun_out/x.err || tail -3 gpurun_out/x.err
python - "$label" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()})
PY
}
B="timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --parity none"
( for x in 0 1 2 3; do
run "normal priority, $x extra streams" SURGE_INGEST_PUSH_PRIORITY=normal SURGE_BENCH_EXTRA_STREAMS=$x $B
run "low priority,    $x extra streams" SURGE_BENCH_EXTRA_STREAMS=$x $B
done ) 2>&1 | tee gpurun_out/r06_e2e_push_priority.txt
