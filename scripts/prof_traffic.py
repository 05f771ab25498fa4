#!/usr/bin/env python3
"""Profile one bench.py configuration with rocprofv3 and emit what profiles/ keeps for it (run on the GPU box):

    python scripts/prof_traffic.py <tag> [bench.py args ...]        e.g.  r02_c3_10M   /   r02_c2 --workload c2

Passes (each its own process, as the MI355X guide prescribes — counters never share a run with a trace):
  1. rocprofv3 --kernel-trace --stats          -> <tag>_kernel_stats.csv   (per-kernel time of the same command)
  2. rocprofv3 --pmc FETCH_SIZE                -> HBM read KB per dispatch (x2 on gfx950: 128-B requests tallied at 64 B)
  3. rocprofv3 --pmc WRITE_SIZE                -> HBM write KB per dispatch
  4. rocprofv3 --pmc SQ_* (two sets)           -> instruction mix / wait / LDS conflicts
Output under gpurun_out/prof_<tag>/: the csv, <tag>_summary.txt, <tag>_bench.json (the JSON line of pass 1) and
<tag>_manifest_entry.json = {kernel, algorithmic_bytes, traffic_bytes, csrc_sha16, source}, the record bench.py's
`roofline.traffic` is served from once merged into profiles/traffic_manifest.json (scripts/merge_manifest.py).
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
sys.path.insert(0, ROOT)


def short_name(k):
    for n in ("fold_rows_kernel", "fold_short_kernel", "fold_sorted_pf_kernel", "fold_sorted_kernel", "fold_chunked_kernel", "fold_tiled_kernel", "relayout_kernel", "fold_slots_tiled_kernel", "fold_slots_kernel", "surge_slots_tiled2", "surge_slots_tiled1", "surge_slots_csr16", "surge_slots_csr8",
              "chunk_stitch_kernel", "stream_probe", "plan_kernel"):
        if n in k:
            return n
    if "surge_v1_flat" in k:  # the flat kernel compiled for the handle's op table (hiprtc)
        return "fold_kernel<FLAT>"
    if "fold_kernel" in k:
        return "fold_kernel<FIXED>" if re.search(r"fold_kernel<0|fold_kernelILi0", k) else "fold_kernel<FLAT>"
    return None


def main():
    tag = sys.argv[1]
    bench_args = sys.argv[2:]
    out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    os.makedirs(out, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")

    def prof(sub, flags):
        d = os.path.join(out, sub)
        log = os.path.join(out, sub + ".log")
        # one pass hung for 25 GPU-minutes once (a FETCH_SIZE pass of the 10 M-aggregate log, rocprofv3 idle after HSA
        # init): every pass gets its own budget and its own process group, and a pass that exceeds it is killed and skipped
        budget = int(os.environ.get("PROF_PASS_TIMEOUT", "240"))
        with open(log, "w") as fh:
            pr = subprocess.Popen(["rocprofv3", "--output-format", "csv"] + flags + ["-d", d, "-o", sub, "--"] + cmd, cwd="/tmp", env=env,
                                  stdout=fh, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                pr.wait(timeout=budget)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, 9)  # exactly the process group started above
                pr.wait()
                fh.write(f"\nPASS KILLED after {budget} s\n")
                print(f"pass {sub}: killed after {budget} s", flush=True)
        return d, log

    d_trace, log_trace = prof("trace", ["--kernel-trace", "--stats"])
    bench_line = None
    for line in open(log_trace):
        if line.startswith("{") and '"metric"' in line:
            bench_line = json.loads(line)
    prof("pmc_fetch", ["--pmc", "FETCH_SIZE"])
    prof("pmc_write", ["--pmc", "WRITE_SIZE"])
    if os.environ.get("PROF_SKIP_SQ") != "1":
        prof("pmc_sq", ["--pmc", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"])
        prof("pmc_sq2", ["--pmc", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM",
                         "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_SALU"])

    lines = ["== kernel stats (rocprofv3 --kernel-trace --stats) of: " + " ".join(cmd[1:]) + " =="]
    for f in glob.glob(os.path.join(d_trace, "**", "*kernel_stats.csv"), recursive=True):
        rows = open(f).read().splitlines()
        lines += rows[:14]
        with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as fh:
            fh.write("\n".join(rows) + "\n")
    avg = defaultdict(dict)
    lines.append("\n== PMC per-dispatch averages (warm-up launches included) ==")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short_name(r.get("Kernel_Name", ""))
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                avg[k][c] = sum(v) / len(v)
                lines.append(f"{sub:10s} {k:22s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
    entry = None
    if bench_line:
        json.dump(bench_line, open(os.path.join(out, f"{tag}_bench.json"), "w"))
        roof = bench_line["roofline"]
        parts = []
        for lbl in roof["kernel"].split("+"):
            name = lbl.split("<")[0].strip()
            if name == "fold_kernel":
                name = "fold_kernel<FLAT>" if "FLAT" in lbl else "fold_kernel<FIXED>"
            parts.append(name)
        fetch = sum(avg.get(p, {}).get("FETCH_SIZE", 0.0) for p in parts)
        write = sum(avg.get(p, {}).get("WRITE_SIZE", 0.0) for p in parts)
        if fetch > 0:
            from bench import csrc_sha16

            traffic = fetch * 1024 * 2 + write * 1024  # KB -> bytes; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
            entry = {"kernel": roof["kernel"], "algorithmic_bytes": roof["algorithmic_bytes"], "traffic_bytes": traffic,
                     "fetch_size_kb": fetch, "write_size_kb": write, "csrc_sha16": csrc_sha16(roof["kernel"]),
                     "kernel_ms_bench": roof["kernel_ms"], "workload": bench_line["config"]["workload"],
                     "source": f"profiles/{tag}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)"}
            json.dump(entry, open(os.path.join(out, f"{tag}_manifest_entry.json"), "w"), indent=1)
            lines.append(f"\n== traffic per launch: FETCH {fetch:.6g} KB x 1024 x 2 + WRITE {write:.6g} KB x 1024 = {traffic:.6g} B; "
                         f"algorithmic {roof['algorithmic_bytes']} B; ratio {traffic / roof['algorithmic_bytes']:.4f} ==")
    open(os.path.join(out, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # keep what travels back small: the raw rocprofv3 trees are scratch
    for sub in ("trace", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        subprocess.run(["rm", "-rf", os.path.join(out, sub)])


if __name__ == "__main__":
    main()
