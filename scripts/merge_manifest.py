#!/usr/bin/env python3
"""Copy gpurun_out/prof_<tag>/<tag>_{summary.txt,kernel_stats.csv,bench.json} into profiles/ and merge the tag's
manifest entry into profiles/traffic_manifest.json (one entry per (kernel, algorithmic_bytes); newest wins)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(ROOT, "profiles")
man_path = os.path.join(prof, "traffic_manifest.json")
man = json.load(open(man_path)) if os.path.exists(man_path) else []
for tag in sys.argv[1:]:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    for suffix in ("summary.txt", "kernel_stats.csv", "bench.json"):
        f = os.path.join(src, f"{tag}_{suffix}")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(prof, f"{tag}_{suffix}"))
    ef = os.path.join(src, f"{tag}_manifest_entry.json")
    if os.path.exists(ef):
        e = json.load(open(ef))
        man = [m for m in man if not (m["kernel"] == e["kernel"] and m["algorithmic_bytes"] == e["algorithmic_bytes"])] + [e]
json.dump(man, open(man_path, "w"), indent=1)
print(f"{len(man)} manifest entries")
