#!/bin/bash
# untouched fields out of the walk / scan (FL_DFLT): the parity suites that fold with the flat kernel, then the numbers
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r4flatspec2; mkdir -p $O
SURGE_TEST_FUZZ_SEEDS=10 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_store.py tests/test_ingest_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.log
grep -n "passed\|failed\|Error" $O/pytest.log | tail -n 5
timeout 300 python scripts/experiments/flat_spec_bench.py > $O/flat_spec.jsonl 2> $O/flat_spec.err; echo "bench rc=$?" | tee -a $O/rc.log
cat $O/flat_spec.jsonl
cd /tmp
for v in counter/1; do
  n=$(echo $v | tr / _)
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_$n -o p -- python /root/repo/scripts/experiments/flat_spec_bench.py --steps 2 --only $v > $O/pmc_$n.log 2>&1; echo "pmc $v rc=$?" | tee -a $O/rc.log
done
cd /root/repo
python - <<'P'
import csv, glob, os, collections
O="gpurun_out/r4flatspec2"
out=[]
for d in sorted(glob.glob(O+"/pmc_*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","")
            if "flat" in k or "fold_kernel" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in acc.items():
        out.append(f"pmc {os.path.basename(d):16s} {k:60s} " + "  ".join(f"{c}={sum(v)/len(v):.5g}" for c,v in sorted(cs.items())))
open(O+"/summary.txt","w").write("\n".join(out)+"\n"); print("\n".join(out))
P
rm -rf $O/pmc_counter_1
