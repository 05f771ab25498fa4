#!/usr/bin/env python3
"""Kernel-level evidence for the bytes -> states path (run on the GPU box):

    python scripts/prof_ingest.py <tag> [bench.py args ...]     e.g.  r04_e2e --workload e2e --steps 6

Every kernel of the command is kept (the ingest runs a dozen small ones), not only the folds scripts/prof_traffic.py
knows by name.  Passes, each its own process (counters never share a run with a trace — MI355X guide, rocprofv3 PMC
slots: SQ 8 per pass, FETCH_SIZE / WRITE_SIZE do not fit one pass):
  1. rocprofv3 --kernel-trace --stats   -> <tag>_kernel_stats.csv: calls, total / average ns per kernel
  2. --pmc SQ set 1                      -> waves, wave cycles, busy cycles, VALU / SALU / LDS instructions, issue stalls
  3. --pmc SQ set 2                      -> parked cycles, LDS bank conflicts, VMEM instructions
  4. --pmc FETCH_SIZE, 5. --pmc WRITE_SIZE -> HBM bytes per dispatch (FETCH x2 on gfx950)
Output under gpurun_out/prof_<tag>/: <tag>_kernel_stats.csv, <tag>_summary.txt (the table below), <tag>_bench.json.
PROF_PASSES=trace,sq1 selects passes (GPU minutes are budgeted).
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "trace": ["--kernel-trace", "--stats"],
    "sq1": ["--pmc", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    "sq2": ["--pmc", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_SCA"],
    "fetch": ["--pmc", "FETCH_SIZE"],
    "write": ["--pmc", "WRITE_SIZE"],
}


def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"\(.*$", "", k)
    k = re.sub(r"rocprim::detail::", "rocprim::", k)
    return k[:70]


def main():
    tag = sys.argv[1]
    bench_args = sys.argv[2:]
    out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    os.makedirs(out, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    want = os.environ.get("PROF_PASSES", "trace,sq1,sq2,fetch,write").split(",")
    budget = int(os.environ.get("PROF_PASS_TIMEOUT", "240"))
    for sub in want:
        d, log = os.path.join(out, sub), os.path.join(out, sub + ".log")
        with open(log, "w") as fh:
            pr = subprocess.Popen(["rocprofv3", "--output-format", "csv"] + PASSES[sub] + ["-d", d, "-o", sub, "--"] + cmd, cwd="/tmp", env=env,
                                  stdout=fh, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                pr.wait(timeout=budget)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, 9)  # exactly the process group started above
                pr.wait()
                fh.write(f"\nPASS KILLED after {budget} s\n")
                print(f"pass {sub}: killed after {budget} s", flush=True)
    lines = ["== " + " ".join(["bench.py"] + bench_args) + " =="]
    bench_line = None
    if "trace" in want:
        for line in open(os.path.join(out, "trace.log")):
            if line.startswith("{") and '"metric"' in line:
                bench_line = json.loads(line)
        for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as fh:
                fh.write(open(f).read())
            lines.append("\n== kernel stats (rocprofv3 --kernel-trace --stats), every kernel above 0.2 % ==")
            lines.append(f"{'kernel':72s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
            for r in rows:
                if float(r["Percentage"]) >= 0.2:
                    lines.append(f"{short(r['Name']):72s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")
    acc = defaultdict(lambda: defaultdict(list))
    for sub in want:
        if sub == "trace":
            continue
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r.get("Kernel_Name", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        lines.append("\n== PMC per-dispatch averages (every dispatch of the run, warm-up included) ==")
        for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
            cs = acc[k]
            lines.append(f"{k}  (n={max(len(v) for v in cs.values())})")
            lines.append("    " + "  ".join(f"{c}={sum(v) / len(v):.5g}" for c, v in sorted(cs.items())))
    if bench_line:
        json.dump(bench_line, open(os.path.join(out, f"{tag}_bench.json"), "w"))
    open(os.path.join(out, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    for sub in want:  # the raw rocprofv3 trees are scratch
        subprocess.run(["rm", "-rf", os.path.join(out, sub)])


if __name__ == "__main__":
    main()
