export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r6d
R=$PWD
cd /tmp
export SHAPE=c3 ROUNDS=1 FOLDS=3 SCHEMAS=builtin VARIANTS="builtin:aot:16,builtin:rtc:16,builtin:rtc:8"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --output-format csv --pmc $pass -d $R/gpurun_out/prof_r6d/$tag -o $tag -- python $R/scripts/lane_spec_ab.py > $R/gpurun_out/prof_r6d/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_r6d/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'sorted' not in k: continue
        k = 'aot16' if 'fold_sorted_pf_kernel' in k else k
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        for c,x in v.items():
            print(f.split('/')[-2], k[:40], c, len(x), sum(x)/len(x))
PY
