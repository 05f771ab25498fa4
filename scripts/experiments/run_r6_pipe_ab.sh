export TMPDIR=/tmp
mkdir -p gpurun_out
run() { tag="$1"; shift; env "$@" timeout 600 python bench.py --workload e2e --steps 60 --warmup 2 --parity none $EXTRA > gpurun_out/x.json 2> gpurun_out/x.err
python - "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/x.json').read().strip().splitlines()[-1]); c=d['config']
    print(sys.argv[1], 'value %.3e'%d['value'], 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'fetch_ms', {k:round(v,2) for k,v in c['fetch_ms'].items()}, 'framing wall', round(c['host_framing_ms_per_fetch'],2), 'recv wall', round(c['receive_copy_ms_per_fetch'],2), 'finish+fold', round(c['finish_and_fold_ms_per_fetch'],2), 'push', round(c['push_async_host_ms_per_fetch'],2))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
}
EXTRA="--framing-threads 3"
run base X=1
run depth5 SURGE_BENCH_DEPTH=5
run depth3 SURGE_BENCH_DEPTH=3
run streams4 SURGE_INGEST_PUSH_STREAMS=4
run streams2 SURGE_INGEST_PUSH_STREAMS=2
run base X=1
EXTRA="--framing-threads 4"
run threads4 X=1
EXTRA="--framing-threads 6"
run threads6 X=1
EXTRA="--framing-threads 3 --two-thread-consumer"
run twothread X=1
