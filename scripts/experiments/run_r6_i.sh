export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "inplace2:--framing-threads 2" "inplace4:--framing-threads 4" "copy12:--framing-by-copy --framing-threads 12" "inplace2b:--framing-threads 2" "inplace3:--framing-threads 3"; do
  tag=${v%%:*}; flags=${v#*:}
  timeout 600 python bench.py --workload e2e --steps 14 --warmup 2 $flags > gpurun_out/r06_e2e_$tag.json 2> gpurun_out/r06_e2e_$tag.err; tail -2 gpurun_out/r06_e2e_$tag.err | cut -c1-300
  python - $tag <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r06_e2e_{sys.argv[1]}.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], {k:(round(c[k],3) if isinstance(c[k],float) else c[k]) for k in ('host_cpu_ms_per_1e6_records','host_cpu_ms_per_1e6_records_without_the_receive_copy','framing_cpu_ms_per_1e6_records','receive_copy_cpu_ms_per_1e6_records','receive_copy_ms_per_fetch','framing_threads','host_framing_ms_per_fetch','finish_and_fold_ms_per_fetch','push_async_host_ms_per_fetch')}, c['fetch_ms'])
PY
done
