export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_e2e
cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_e2e -o e2e -- python $R/bench.py --workload e2e --steps 10 --warmup 2 --parity none > /tmp/prof_e2e.log 2>&1
tail -3 /tmp/prof_e2e.log | cut -c1-400
ls -la /tmp/prof_e2e/ | head -20
for f in /tmp/prof_e2e/*stats.csv; do echo "== $f"; head -40 $f | cut -c1-260; cp $f $R/gpurun_out/prof_e2e/; done
python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('/tmp/prof_e2e/*memory_copy_trace.csv'):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    agg=collections.Counter(); cnt=collections.Counter()
    sizes=collections.defaultdict(list)
    for r in rows:
        k=(r.get('Direction') or r.get('Name'))
        d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        agg[k]+=d; cnt[k]+=1
        sizes[k].append(int(r.get('Bytes',0) or 0))
    for k in agg: print(k, cnt[k], 'total ms %.2f'%(agg[k]/1e6), 'avg us %.1f'%(agg[k]/cnt[k]/1e3), 'bytes total %.1f MB'%(sum(sizes[k])/1e6))
PY
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/prof_e2e/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(rows[0].keys())
# the window of one steady push: between the 9th and 10th crc_kernel
crc=[i for i,r in enumerate(rows) if 'crc_kernel' in r['Kernel_Name']]
a,b=crc[8],crc[9]
t0=int(rows[a]['Start_Timestamp'])
def short(n):
    for k in ('copyBuffer','fillBuffer','crc_kernel','lz4_parse','lz4_exec','section_kernel','probe_kernel','flag_kernel','assign_kernel','finalize_kernel','surge_v1_flat','groupby','plan_dev','merge_sort','radix_sort','scan','lookback','rollback','rekey','chain','records_kernel'):
        if k in n: return k
    return n[:40]
print("one fetch's window, in start order: kernel, start us (relative), duration us, grid, queue")
for r in rows[a-12:b-12]:
    print('%-16s %9.1f %8.1f grid %-9s wg %-5s q %s'%(short(r['Kernel_Name']),(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Grid_Size_X') or r.get('Grid_Size'),r.get('Workgroup_Size_X') or r.get('Workgroup_Size'),r.get('Queue_Id')))
cb=[r for r in rows if 'copyBuffer' in r['Kernel_Name']]
by=collections.defaultdict(list)
for r in cb: by[r.get('Grid_Size_X') or r.get('Grid_Size')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('copyBuffer by grid size: grid -> n, avg us')
for g,v in sorted(by.items(), key=lambda kv:-sum(kv[1])): print(g, len(v), round(sum(v)/len(v),1), 'total', round(sum(v),1))
PY
