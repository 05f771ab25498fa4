#!/bin/bash
# the flat kernel compiled per v1 schema: the GPU suite (it is the default build of every flat fold now), then the
# before / after numbers with a kernel trace and the VALU counter
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r4flatspec; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.log
tail -n 4 $O/pytest.log
timeout 300 python scripts/experiments/flat_spec_bench.py > $O/flat_spec.jsonl 2> $O/flat_spec.err; echo "bench rc=$?" | tee -a $O/rc.log
cat $O/flat_spec.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python /root/repo/scripts/experiments/flat_spec_bench.py --steps 5 > $O/trace.log 2>&1; echo "trace rc=$?" | tee -a $O/rc.log
for v in counter/1 counter/0 default/1 default/0; do
  n=$(echo $v | tr / _)
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_$n -o p -- python /root/repo/scripts/experiments/flat_spec_bench.py --steps 2 --only $v > $O/pmc_$n.log 2>&1; echo "pmc $v rc=$?" | tee -a $O/rc.log
done
cd /root/repo
python - <<'P'
import csv, glob, os, collections
O="gpurun_out/r4flatspec"
out=[]
for f in glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flat" in r["Name"] or "fold_kernel" in r["Name"]:
            out.append(f"trace  {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
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
rm -rf $O/trace $O/pmc_counter_1 $O/pmc_counter_0 $O/pmc_default_1 $O/pmc_default_0
