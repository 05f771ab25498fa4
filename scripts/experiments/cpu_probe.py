"""What the GPU box's host really gives the cpu_baseline leg: logical CPUs, affinity, cgroup quota, and the oracle's rate by thread count."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle import oracle
from surge_amd import synth

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
try:
    print(open("/proc/cpuinfo").read().count("processor\t"), "processors;", [l for l in open("/proc/cpuinfo") if "model name" in l][0].strip())
    print("loadavg", open("/proc/loadavg").read().strip())
except OSError:
    pass
lens = synth.zipf_lengths(np.arange(140_000, dtype=np.int64), 3)
so, ev = synth.csr_log(lens, 3)
n = int(so[-1])
print("sample events", n)
for th in (1, 8, 16, 32, 64, 128, 256):
    oracle.fold_csr_repeated(so, ev, th, 1)
    reps = 3 if th == 1 else 20
    t0 = time.perf_counter()
    oracle.fold_csr_repeated(so, ev, th, reps)
    dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {n * reps / dt / 1e9:.3f} G events/s")
