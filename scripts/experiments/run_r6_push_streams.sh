export TMPDIR=/tmp
mkdir -p gpurun_out
run() { label=$1; shift
env "$@" > gpurun_out/x.json 2> gpurun_out/x.err || tail -3 gpurun_out/x.err
python - "$label" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'main', c['host_cpu_ms_per_1e6_records_by_thread'].get('MainThread'), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()}, 'finish+fold wall', round(c['finish_and_fold_ms_per_fetch'],2), 'push wall', round(c['push_async_host_ms_per_fetch'],2), 'framing wall', round(c['host_framing_ms_per_fetch'],2), 'recv wall', round(c['receive_copy_ms_per_fetch'],2))
PY
}
B="timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --parity none"
for rep in 1 2; do
run "streams 2, depth 4, 2 frm" SURGE_INGEST_PUSH_STREAMS=2 $B
run "streams 3, depth 4, 2 frm" SURGE_INGEST_PUSH_STREAMS=3 $B
run "streams 3, depth 4, 3 frm" SURGE_INGEST_PUSH_STREAMS=3 $B --framing-threads 3
run "streams 3, depth 5, 3 frm" SURGE_INGEST_PUSH_STREAMS=3 SURGE_BENCH_DEPTH=5 $B --framing-threads 3
done 2>&1 | tee gpurun_out/r06_e2e_push_streams.txt
