export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_rehearsal.py -x -q -m gpu -k "mixed or shards" --timeout 300 2>&1 | tail -5
timeout 600 python bench.py --workload e2e --e2e-topic mixed --steps 100 --warmup 2 > gpurun_out/r06_e2e_mixed.json 2> gpurun_out/r06_e2e_mixed.err; tail -2 gpurun_out/r06_e2e_mixed.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e2e_mixed.json').read().strip().splitlines()[-1]); c=d['config']
print('mixed value %.3e'%d['value'], 'parity', d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], 'fetches', c['fetches'], 'wire B/rec', round(c['wire_bytes_per_record'],1), 'cpu', round(c['host_cpu_ms_per_1e6_records'],2), 'by thread', c['host_cpu_ms_per_1e6_records_by_thread'])
print(' fetch_ms', c['fetch_ms'], 'framing ms', round(c['host_framing_ms_per_fetch'],3), 'finish+fold', round(c['finish_and_fold_ms_per_fetch'],3), 'decoder', c['decoder'])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mixed -o mixed -- python $OLDPWD/bench.py --workload e2e --e2e-topic mixed --steps 100 --warmup 2 --parity none > /tmp/prof_mixed.log 2>&1
tail -1 /tmp/prof_mixed.log | cut -c1-200
cp /tmp/prof_mixed/*kernel_stats.csv $OLDPWD/gpurun_out/r06_e2e_mixed_kernel_stats.csv
head -14 $OLDPWD/gpurun_out/r06_e2e_mixed_kernel_stats.csv | cut -c1-180
