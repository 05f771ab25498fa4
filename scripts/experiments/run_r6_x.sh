export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
for topic in counter mixed; do
rm -rf /tmp/prof_d1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d1 -o d1 -- python $R/bench.py --workload e2e --e2e-topic $topic --steps 20 --warmup 2 --parity none --serial-framing > /tmp/prof_d1.log 2>&1
tail -1 /tmp/prof_d1.log | cut -c1-120
cp /tmp/prof_d1/*kernel_stats.csv $R/gpurun_out/r06_e2e_${topic}_depth1_kernel_stats.csv
python3 - $R/gpurun_out/r06_e2e_${topic}_depth1_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %5s avg us %9.1f  total ms %8.2f'%(r['Name'][:60].replace('(anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, int(r['TotalDurationNs'])/1e6))
PY
done
