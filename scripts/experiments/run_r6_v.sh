export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for t in 3 2; do
timeout 600 python bench.py --workload e2e --steps 100 --warmup 2 --framing-threads $t > gpurun_out/r06_e2e_threads_cpu_$t.json 2> gpurun_out/r06_e2e_threads_cpu_$t.err
python - $t <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r06_e2e_threads_cpu_{sys.argv[1]}.json').read().strip().splitlines()[-1]); c=d['config']
print('threads', sys.argv[1], 'value %.3e'%d['value'], 'parity', c.get('parity'), 'cpu', round(c['host_cpu_ms_per_1e6_records'],3), 'w/o recv', round(c['host_cpu_ms_per_1e6_records_without_the_receive_copy'],3))
print('   by thread', c['host_cpu_ms_per_1e6_records_by_thread'])
PY
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -o e2e -- python $R/bench.py --workload e2e --steps 30 --warmup 2 --parity none > /tmp/prof_e2e.log 2>&1
tail -1 /tmp/prof_e2e.log | cut -c1-300
cp /tmp/prof_e2e/*kernel_stats.csv $R/gpurun_out/r06_e2e_slicing_crc_kernel_stats.csv
head -30 $R/gpurun_out/r06_e2e_slicing_crc_kernel_stats.csv | cut -c1-200
