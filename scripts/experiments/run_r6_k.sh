export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  tag=$1; shift
  timeout 600 python bench.py --workload e2e --steps 14 --warmup 2 "$@" > gpurun_out/r06_e2e_$tag.json 2> gpurun_out/r06_e2e_$tag.err; tail -2 gpurun_out/r06_e2e_$tag.err | cut -c1-300
  python - $tag <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r06_e2e_{sys.argv[1]}.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], {k:(round(c[k],3) if isinstance(c[k],float) else c[k]) for k in ('host_cpu_ms_per_1e6_records','host_cpu_ms_per_1e6_records_without_the_receive_copy','framing_cpu_ms_per_1e6_records','receive_copy_cpu_ms_per_1e6_records','receive_copy_ms_per_fetch','framing_threads','host_framing_ms_per_fetch','finish_and_fold_ms_per_fetch','push_async_host_ms_per_fetch')}, c['fetch_ms'])
PY
}
run inplace2 --framing-threads 2
SURGE_INGEST_WAIT=block run inplace2_block --framing-threads 2
run inplace3 --framing-threads 3
SURGE_INGEST_WAIT=block run inplace3_block --framing-threads 3
run copy12 --framing-by-copy --framing-threads 12
SURGE_INGEST_WAIT=block run copy12_block --framing-by-copy --framing-threads 12
run inplace2_two --framing-threads 2 --two-thread-consumer
