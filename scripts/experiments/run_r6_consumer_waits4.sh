export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, waits, env...
label=$1; w=$2; shift 2
env "$@" timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --framing-threads 2 --consumer-waits $w --parity none > gpurun_out/x.json 2> gpurun_out/x.err || tail -3 gpurun_out/x.err
python - "$label" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
cc=c['consumer_cpu_ms_per_fetch']
print(sys.argv[1], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'main', c['host_cpu_ms_per_1e6_records_by_thread'].get('MainThread'), 'recv wall', round(c['receive_copy_ms_per_fetch'],2), 'framing wall', round(c['host_framing_ms_per_fetch'],2), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()}, 'finish+fold wall', round(c['finish_and_fold_ms_per_fetch'],2))
PY
}
for rep in 1 2; do
run "finish streams2 own      " finish SURGE_INGEST_PUSH_STREAMS=2
run "finish streams2 null     " finish SURGE_INGEST_PUSH_STREAMS=2 SURGE_BENCH_OWN_STREAM=0
run "finish streams2 own hwq8 " finish SURGE_INGEST_PUSH_STREAMS=2 GPU_MAX_HW_QUEUES=8
run "finish streams1 own      " finish SURGE_INGEST_PUSH_STREAMS=1
run "both streams2            " both SURGE_INGEST_PUSH_STREAMS=2
done 2>&1 | tee gpurun_out/r06_e2e_consumer_waits_queues2.txt
