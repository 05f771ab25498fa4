"""-m gpu: BASELINE.json's full-size configs, checked through size-independent properties.

C2 = 1 M aggregates x 256 events (4.1 GB), C3 = 10 M aggregates with Zipf(1..4096) event counts
(~4.6e9 events, ~74 GB, generated on the GPU).  EVERY aggregate of C2, C3 and the C4 shard is folded by the CPU
oracle too (the log is streamed to the host in 4 GiB slices of whole aggregates and folded on all host threads:
bench.full_log_parity, ~1-2 min for the 74 GB log) and compared byte for byte; beside that: every kernel family agrees
bit for bit on the whole log, replay is idempotent, and — for a Counter-only log, where the fold is linear — every
field equals an independent torch segment reduction (prefix sums over the raw events).

Set SURGE_TEST_C3_AGGREGATES to shrink C3 (default 10_000_000).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (full_log_parity: the whole-log oracle check bench.py's cpu_baseline leg runs)

from oracle import oracle
from surge_amd import schema as S
from surge_amd import synth
from surge_amd.replay import ReplayEngine

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def fold_all(so, ev, algos):
    out = {}
    with ReplayEngine() as eng:
        n = so.numel() - 1
        buf = torch.empty((n, 64), dtype=torch.uint8, device=DEV)
        eng.load_csr(so, ev, None, buf)
        for a in algos:
            eng.fold(a)
            eng.synchronize()
            assert eng.stats().last_algo == a
            out[a] = buf.clone()
        eng.fold(algos[0])  # idempotence: replay is a pure function of the log
        eng.synchronize()
        assert torch.equal(buf, out[algos[0]])
    return out


def whole_log_parity(states, so, ev):
    cores, _, _ = bench.effective_cpus()
    r = bench.full_log_parity(so, ev, states, cores)
    print(f"full-log oracle parity: {r['aggregates_checked']} aggregates / {r['events_checked']} events in {r['seconds']:.1f} s "
          f"({r['cpu_fold_seconds']:.1f} s of it the CPU fold on {r['threads']} threads, {r['slices']} slices)")
    assert r["aggregates_checked"] == so.numel() - 1 and r["events_checked"] == int(so[-1]) - int(so[0])
    assert r["mismatching_aggregates"] == 0, f"{r['mismatching_aggregates']} aggregates differ from the oracle, first {r['first_mismatch']}"


def slice_parity(states, so, ev, a0, a1):
    e0, e1 = int(so[a0]), int(so[a1])
    sub_off = (so[a0:a1 + 1] - so[a0]).cpu().numpy()
    exp = oracle.fold_csr(sub_off, synth.to_event_records(ev[e0:e1]))
    assert states[a0:a1].cpu().numpy().tobytes() == exp.tobytes()


def counter_reference(so, ev):
    """Independent torch restatement for INC/DEC/NOOP-only logs: the fold is linear."""
    w0, w1 = ev[:, 0], ev[:, 1]
    ty = w0 & 0xFFFFFFFF
    seq = w0 >> 32
    arg = ((w1 & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000  # sign-extend the low word
    upd = (ty == S.EVT_INC) | (ty == S.EVT_DEC)
    signed = torch.where(ty == S.EVT_INC, arg, torch.where(ty == S.EVT_DEC, -arg, torch.zeros_like(arg)))
    zero = torch.zeros(1, dtype=torch.int64, device=ev.device)
    csum = torch.cat([zero, torch.cumsum(signed, 0)])
    sum64 = csum[so[1:]] - csum[so[:-1]]
    cnt = torch.cat([zero, torch.cumsum(upd.to(torch.int64), 0)])
    evc = cnt[so[1:]] - cnt[so[:-1]]
    # version = seq of the last INC/DEC event of the segment (0 if none)
    idx = torch.arange(ev.shape[0], dtype=torch.int64, device=ev.device)
    last_upd = torch.cummax(torch.where(upd, idx, torch.full_like(idx, -1)), 0).values
    last_in_seg = last_upd[so[1:] - 1]
    has = last_in_seg >= so[:-1]
    version = torch.where(has, seq[last_in_seg.clamp(min=0)], torch.zeros_like(last_in_seg))
    return sum64, evc, version


def check_counter_fields(states, so, ev, max_chunk_events=300_000_000):
    """Chunked over aggregate ranges so the int64 temporaries stay far below HBM capacity."""
    n = so.numel() - 1
    a0 = 0
    while a0 < n:
        limit = int(so[a0]) + max_chunk_events
        a1 = int(torch.searchsorted(so, torch.tensor([limit], device=so.device), right=True)[0]) - 1
        a1 = max(a0 + 1, min(a1, n))
        e0, e1 = int(so[a0]), int(so[a1])
        sub_so = so[a0:a1 + 1] - so[a0]
        st = states[a0:a1].view(torch.int32).reshape(-1, 16)
        sum64, evc, version = counter_reference(sub_so, ev[e0:e1])
        count32 = ((sum64 & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000  # wrap like a JVM Int
        assert torch.equal(st[:, 0].to(torch.int64), count32)
        assert torch.equal(st[:, 1].to(torch.int64), version)
        got_sum = (st[:, 2].to(torch.int64) & 0xFFFFFFFF) | (st[:, 3].to(torch.int64) << 32)
        assert torch.equal(got_sum, sum64)
        assert torch.equal(st[:, 8].to(torch.int64) & 0xFFFFFFFF, evc)
        assert bool((st[:, 9] == S.STATE_PRESENT).all())  # every aggregate has >= 1 event and materialises
        del sum64, evc, version, st
        a0 = a1


def test_c2_full_size_1m_aggregates_x_256_events():
    A, L = 1_000_000, 256
    so, ev = synth.fixed_log_device(A, L, 2, DEV)
    res = fold_all(so, ev, [S.ALGO_ROWS, S.ALGO_FIXED, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_TILED])
    for a in (S.ALGO_FIXED, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_TILED):
        assert torch.equal(res[S.ALGO_ROWS], res[a]), f"algo {a} differs from rows on the full C2 log"
    whole_log_parity(res[S.ALGO_ROWS], so, ev)
    slice_parity(res[S.ALGO_ROWS], so, ev, 0, 20_000)
    slice_parity(res[S.ALGO_ROWS], so, ev, A - 5_000, A)
    del res
    # linearity on a Counter-only log of the same shape
    so, ev = synth.fixed_log_device(A, L, 3, DEV, mix=synth.C1_MIX)
    res = fold_all(so, ev, [S.ALGO_ROWS, S.ALGO_FLAT])
    assert torch.equal(res[S.ALGO_ROWS], res[S.ALGO_FLAT])
    check_counter_fields(res[S.ALGO_ROWS], so, ev)


def test_c3_full_size_10m_aggregates_zipf():
    A = int(os.environ.get("SURGE_TEST_C3_AGGREGATES", "10000000"))
    lens = synth.zipf_lengths(torch.arange(A, dtype=torch.int64, device=DEV), 3)
    # C3 type mix: oracle slices + agreement between the linear-stream and the sorted-rows kernels
    so, ev = synth.csr_log_device(lens, 3)
    res = fold_all(so, ev, [S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED])
    assert torch.equal(res[S.ALGO_FLAT], res[S.ALGO_SORTED]), "flat and sorted-rows differ on the full C3 log"
    assert torch.equal(res[S.ALGO_FLAT], res[S.ALGO_CHUNKED]), "flat and chunked-rows differ on the full C3 log"
    assert torch.equal(res[S.ALGO_FLAT], res[S.ALGO_TILED]), "flat and tile-major differ on the full C3 log"
    whole_log_parity(res[S.ALGO_TILED], so, ev)  # all 10 M aggregates, all 4.6e9 events, byte for byte
    slice_parity(res[S.ALGO_FLAT], so, ev, 0, 4_000)
    slice_parity(res[S.ALGO_FLAT], so, ev, A - 3_000, A)
    slice_parity(res[S.ALGO_FLAT], so, ev, A // 2, A // 2 + 3_000)
    del res, ev
    torch.cuda.empty_cache()
    # Counter-only log of the same shape: every field against the independent segment reduction
    so, ev = synth.csr_log_device(lens, 4, mix=synth.C1_MIX)
    with ReplayEngine() as eng:
        buf = torch.empty((A, 64), dtype=torch.uint8, device=DEV)
        eng.load_csr(so, ev, None, buf)
        eng.fold()
        eng.synchronize()
        # AUTO: sorted rows once the chunk target (bytes / 4 MB) reaches the longest aggregate (~2.2 M Zipf(1..4096)
        # aggregates = 16.4 GB), chunked rows from ~1.5 GB (0.2 M aggregates), FLAT below
        if A >= 2_500_000:
            assert eng.stats().last_algo == S.ALGO_SORTED
        elif 250_000 <= A <= 2_000_000:
            assert eng.stats().last_algo == S.ALGO_CHUNKED
        elif A <= 150_000:
            assert eng.stats().last_algo == S.ALGO_FLAT
    check_counter_fields(buf, so, ev)


def test_c4_shard_size_1_25m_aggregates_zipf_every_kernel_agrees():
    # what ONE GPU of the 8-GPU config holds (BASELINE C4: 10 M Zipf aggregates / 8): AUTO = chunked rows here
    A = 1_250_000
    lens = synth.zipf_lengths(torch.arange(A, dtype=torch.int64, device=DEV), 3)
    so, ev = synth.csr_log_device(lens, 3, mix=synth.STRESS_MIX)  # tombstones, throws, REQUIRE runs across chunk cuts
    res = fold_all(so, ev, [S.ALGO_CHUNKED, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_TILED])
    assert torch.equal(res[S.ALGO_CHUNKED], res[S.ALGO_FLAT]) and torch.equal(res[S.ALGO_CHUNKED], res[S.ALGO_SORTED])
    assert torch.equal(res[S.ALGO_CHUNKED], res[S.ALGO_TILED])
    whole_log_parity(res[S.ALGO_TILED], so, ev)  # every aggregate of the shard, stress mix
    slice_parity(res[S.ALGO_CHUNKED], so, ev, 0, 6_000)
    slice_parity(res[S.ALGO_CHUNKED], so, ev, A - 4_000, A)
    with ReplayEngine() as eng:
        buf = torch.empty((A, 64), dtype=torch.uint8, device=DEV)
        eng.load_csr(so, ev, None, buf)
        eng.fold()
        eng.synchronize()
        assert eng.stats().last_algo == S.ALGO_CHUNKED
        assert torch.equal(buf, res[S.ALGO_CHUNKED])
