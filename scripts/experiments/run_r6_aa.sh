export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
rm -rf /tmp/prof_d1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_d1 -o d1 -- python $R/bench.py --workload e2e --steps 8 --warmup 2 --parity none --serial-framing > /tmp/prof_d1.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_d1/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sec=[r for r in rows if 'section_kernel' in r['Kernel_Name'] or 'lz4_parse' in r['Kernel_Name'] or 'lz4_exec' in r['Kernel_Name']]
for r in sec[-36:]:
    print(r['Kernel_Name'][:50].replace('(anonymous namespace)::',''), 'grid', r.get('Grid_Size_X'), 'wg', r.get('Workgroup_Size_X'), 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), 'sgpr', r.get('SGPR_Count'), 'scratch', r.get('Scratch_Size'), 'us %.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
