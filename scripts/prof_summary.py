"""Summarise rocprofv3 output dirs produced by scripts/prof.sh into one text file."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]

def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for i, line in enumerate(fh):
            if i < 12:
                print(line.rstrip())

print("\n== PMC per-dispatch averages for fold kernels ==")
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(os.path.join(sub, "**", "*counter_collection.csv")):
        k = r.get("Kernel_Name", "")
        if "fold_" not in k and "stream_probe" not in k:
            continue
        short = "fold_sorted" if "fold_sorted" in k else "fold_rows" if "fold_rows" in k else ("fold_kernel<FIXED>" if "ILi0" in k or "<0>" in k else ("fold_kernel<FLAT>" if "fold_kernel" in k else "stream_probe"))
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{sub:10s} {k:22s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
