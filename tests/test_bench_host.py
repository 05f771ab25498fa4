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


def test_mixed_topic_bank_account_events_with_headers_decode_to_the_generators_source_events():
    """bench.py --e2e-topic mixed on the host: the generator's BankAccount records (UUID keys without ':', Double balances as
    play-json text — a sample compared with the fixture's event writer byte for byte), written by the independent writer with
    two headers per record into transactional lz4 batches, framed and decoded by the library's HOST decoder: per account the
    decoded events are the generator's source events in order, and the oracle's fold of either is the same state."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import topic_gen
    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.ingest import EventsTopicIngest

    topic = bench.MixedTopic(np, synth, S)
    A, P = 3000, 8
    part = topic.partitions(A, P, None, None, None)
    assert part.min() >= 0 and part.max() < P and np.unique(part).shape[0] == P
    ids = np.arange(A, dtype=np.int64)
    counts = topic.counts(ids, 8)
    long_rows = counts >= topic.LONG_MIN
    assert 5 <= int(long_rows.sum()) <= 60 and counts[long_rows].max() <= 256 and counts[~long_rows].max() <= 8
    a = np.repeat(ids, counts)
    j = (np.arange(a.shape[0]) - np.repeat(np.cumsum(counts) - counts, counts) + 1).astype(np.int32)
    order = np.argsort(j, kind="stable")  # rounds: the j-th event of every account that has one
    a, j = a[order], j[order]
    tmpl = topic.model.event_json_template()
    hdr = topic_gen.set_record_headers(topic.headers)
    try:
        assert hdr > 80
        handles = [EventsTopicIngest() for _ in range(P)]
        got = {}
        with topic_gen.WireTopic(P, 64, 16384, "lz4", 5, 4) as w:
            for lo in range(0, a.shape[0], 5000):
                aa, jj = a[lo:lo + 5000], j[lo:lo + 5000]
                k, ko, v, vo = topic.records(topic_gen, aa, jj)
                for i in range(0, aa.shape[0], 97):
                    topic.check_sample(aa, jj, i, bytes(k[ko[i]:ko[i + 1]]), bytes(v[vo[i]:vo[i + 1]]))
                last = lo + 5000 >= a.shape[0]
                for q, data in enumerate(w.fetch(part[aa], k, ko, v, vo, last=last)):
                    if data:
                        assert b"traceparent" not in data[:61]  # (lz4: the header text is inside the compressed records)
                        handles[q].feed(data)
                        agg, ev, _ = handles[q].drain_json(tmpl)
                        keys = handles[q].key_table().keys
                        for x, e in zip(agg, ev):
                            got.setdefault(keys[int(x)], []).append((int(e["type"]), int(e["seq"]), int(e["raw"])))
            assert w.counts["records_aborted"] > 0 and w.counts["control_batches"] > 0
        for h in handles:
            h.close()
    finally:
        topic_gen.set_record_headers(())
    assert len(got) == A
    kb = np.frombuffer("".join(got).encode(), np.uint8)
    key_ids = topic.ids_of_keys(kb, np.arange(A + 1, dtype=np.int64) * 36, A)
    assert np.array_equal(np.sort(key_ids), ids)
    off = np.zeros(A + 1, np.int64)
    np.cumsum(counts[key_ids], out=off[1:])
    src = topic.source_events(np.repeat(key_ids, counts[key_ids]), (np.arange(off[-1]) - np.repeat(off[:-1], counts[key_ids]) + 1).astype(np.int32))
    dec = np.zeros(off[-1], dtype=S.EVENT_DTYPE)
    flat = [e for key in got for e in got[key]]
    dec["type"], dec["seq"], dec["raw"] = [x[0] for x in flat], [x[1] for x in flat], np.array([x[2] for x in flat], np.uint64)
    assert dec.tobytes() == src.tobytes()
    exp = oracle.fold_csr(off, src, None, topic.model.event_algebra())
    assert float(exp["balance"][0]) == float(np.array([src["raw"][off[1] - 1]], np.uint64).view(np.float64)[0])  # the last balance written
