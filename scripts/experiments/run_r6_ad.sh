export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
export SURGE_REPLAY_LIB=$R/surge_amd/libsurge_replay_exp.so
for mode in "2 lane 64" "2 lane 128" "2 lane 192" "2 lane 256" "2 wave 64" "2 wave 192" "1 lane 64" "1 lane 192" "0 lane 64" "0 wave 64" "0 lane 128" "0 wave 128"; do
set -- $mode
rm -rf /tmp/prof_d1
SURGE_DBG_DECODE=$1 SURGE_INGEST_CHAIN=$2 SURGE_INGEST_SEC_THREADS=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d1 -o d1 -- python $R/bench.py --workload e2e --steps 12 --warmup 2 --parity none --serial-framing > /tmp/prof_d1.log 2>&1
python3 - "$mode" <<'PY'
import csv,sys,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/prof_d1/*kernel_stats.csv')[0])))
for r in rows:
    if 'section_kernel' in r['Name']:
        print(sys.argv[1], '%-30s calls %5s avg us %9.1f  total ms %8.2f max %8.1f'%(r['Name'][:30].replace('(anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, int(r['TotalDurationNs'])/1e6, float(r['MaxNs'])/1e3))
PY
done
