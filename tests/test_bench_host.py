"""Host-side pieces of bench.py that need no GPU: the effective CPU count behind `cpu_baseline.cores`, and the link
between `roofline.traffic` and the rocprofv3 PMC passes committed under profiles/."""
import builtins
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _with_files(monkeypatch, files):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)


def test_effective_cpus_is_the_affinity_capped_by_the_cgroup_quota(monkeypatch):
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"})   # the GPU boxes of this project
    assert bench.effective_cpus() == (16, 256, 16.0)
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "max 100000\n"})       # cgroup v2, unlimited
    assert bench.effective_cpus() == (256, 256, None)
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "250000 100000\n"})     # fractional quota rounds up
    assert bench.effective_cpus() == (3, 256, 2.5)
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})  # cgroup v1
    assert bench.effective_cpus() == (8, 256, 8.0)
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})
    assert bench.effective_cpus() == (256, 256, None)
    _with_files(monkeypatch, {})                                                # no cgroup files at all
    assert bench.effective_cpus() == (256, 256, None)


def test_traffic_manifest_is_keyed_by_kernel_bytes_and_source_hash(tmp_path, monkeypatch):
    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic_manifest.json")))
    assert entries and all({"kernel", "algorithmic_bytes", "traffic_bytes", "csrc_sha16", "source"} <= set(e) for e in entries)
    for e in entries:  # what the counters saw is never less than ~the algorithmic bytes and never wildly more
        assert 0.95 < e["traffic_bytes"] / e["algorithmic_bytes"] < 1.25, e["kernel"]
        assert os.path.exists(os.path.join(ROOT, e["source"].split(" ")[0])), e["source"]
    current = [e for e in entries if e["csrc_sha16"] == bench.csrc_sha16(e["kernel"])]  # (per kernel: the sources THAT kernel is built from)
    e = (current or entries)[0]
    got, src = bench.pmc_traffic(e["kernel"], e["algorithmic_bytes"])
    if current:
        assert got == e["traffic_bytes"] and src == e["source"]
    else:
        assert got is None and src.startswith("stale")
    # another shape of the same kernel has no committed measurement: null, not a guess
    assert bench.pmc_traffic(e["kernel"], e["algorithmic_bytes"] + 16) == (None, None)
    # a kernel source edit invalidates every entry (the bench line then says "stale" instead of quoting old counters)
    assert bench.csrc_sha16("fold_rows_kernel<8>") != bench.csrc_sha16("fold_chunked_kernel<16> + chunk_stitch_kernel") != bench.csrc_sha16("surge_slots_tiled2")
    monkeypatch.setattr(bench, "csrc_sha16", lambda kernel="": "0" * 16)
    got, src = bench.pmc_traffic(e["kernel"], e["algorithmic_bytes"])
    assert got is None and src.startswith("stale")


def test_committed_profiles_describe_the_current_kernel_sources():
    """A reminder rather than a gate: when a fold kernel changes, scripts/prof_traffic.py has to run again on a GPU."""
    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic_manifest.json")))
    stale = sorted({e["source"].split(" ")[0] for e in entries if e["csrc_sha16"] != bench.csrc_sha16(e["kernel"])})
    if stale:
        pytest.skip("kernel sources changed since these PMC passes were taken (bench.py reports traffic: null for them): " + ", ".join(stale))


def test_algo_names_and_plain_multi_gpu_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must re-exec under torch.distributed.run — the way the driver
    starts the N > 1 bench.  Without a GPU the ranks then stop at the no-CPU-fallback check, which is what shows they ran."""
    import subprocess

    import torch

    assert bench.parse_algo("tiled") == 7 and bench.parse_algo("AUTO") == 0 and bench.parse_algo("5") == 5 and bench.parse_algo(None) is None
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_bench_rehearsal.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert res.returncode != 0
    # both ranks were started by torch.distributed.run (its failure report lists them); the first to reach the check stops the job
    assert res.stderr.count("bench.py needs a GPU") >= 1 and "local_rank: 1" in res.stderr and "local_rank: 0" in res.stderr, res.stderr[-3000:]
