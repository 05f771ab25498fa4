"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bit-exact is the bar: all state fields are integers or bit-copied f64 payloads.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from golden_util import NAMES, load_golden
from oracle import oracle
from surge_amd import schema as S
from surge_amd import synth
from fixture_models import BANK_ACCOUNT_ALGEBRA, COUNTER_ALGEBRA
from surge_amd.replay import ReplayEngine, ReplayError

pytestmark = pytest.mark.gpu


def gpu_fold(seg_off, events, init=None, algebra=S.DEFAULT_ALGEBRA, algo=S.ALGO_AUTO):
    with ReplayEngine(algebra) as eng:
        eng.load_csr(seg_off, events, init)
        eng.fold(algo)
        out = eng.snapshot()
        st = eng.stats()
    return out, st


def assert_same(got, exp, seg_off):
    if got.tobytes() != exp.tobytes():
        bad = np.nonzero(got != exp)[0]
        a = bad[0]
        raise AssertionError(
            f"{bad.size} aggregates differ; first {a} (len {seg_off[a + 1] - seg_off[a]}):\n got {got[a]}\n exp {exp[a]}"
        )


@pytest.mark.parametrize("name", NAMES)
def test_golden_vectors(name):
    seg_off, events, init, expected = load_golden(name)
    got, _ = gpu_fold(seg_off, events, init)
    assert_same(got, expected, seg_off)


@pytest.mark.parametrize(
    "n_agg,length,mix",
    [(1000, 100, synth.C1_MIX), (4096, 256, synth.C2_MIX), (333, 48, synth.STRESS_MIX), (100, 4096, synth.STRESS_MIX),
     (7, 16, synth.STRESS_MIX), (1, 1024, synth.C2_MIX), (5000, 1, synth.STRESS_MIX), (64, 1040, synth.C2_MIX)],
)
def test_fixed_fan_in(n_agg, length, mix):
    so, ev = synth.fixed_log(n_agg, length, seed=n_agg + length, mix=mix)
    exp = oracle.fold_csr(so, ev)
    got, st = gpu_fold(so, ev)
    assert_same(got, exp, so)
    assert st.last_algo == (S.ALGO_FIXED if length % 16 == 0 else S.ALGO_FLAT)  # too few aggregates for ROWS
    got_flat, st = gpu_fold(so, ev, algo=S.ALGO_FLAT)
    assert st.last_algo == S.ALGO_FLAT
    assert_same(got_flat, exp, so)
    got_sorted, st = gpu_fold(so, ev, algo=S.ALGO_SORTED)
    assert st.last_algo == S.ALGO_SORTED
    assert_same(got_sorted, exp, so)
    if length % 16 == 0:
        got_rows, st = gpu_fold(so, ev, algo=S.ALGO_ROWS)
        assert st.last_algo == S.ALGO_ROWS
        assert_same(got_rows, exp, so)


@pytest.mark.parametrize("seed,mix", [(3, synth.C2_MIX), (4, synth.STRESS_MIX)])
def test_zipf_csr(seed, mix):
    so, ev = synth.zipf_log(20000, seed, mix=mix)
    exp = oracle.fold_csr(so, ev)
    got, st = gpu_fold(so, ev)
    assert st.last_algo == S.ALGO_FLAT  # 150 MB: below the ~1.5 GB where one lane per chunk starts to pay
    assert_same(got, exp, so)
    for algo in (S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED):
        got, st = gpu_fold(so, ev, algo=algo)
        assert st.last_algo == algo
        assert_same(got, exp, so)


@pytest.mark.parametrize("chunk_t", [16, 64, 256, 1016])
def test_chunked_rows_resolve_presence_across_chunks(chunk_t, monkeypatch):
    # K2c: an aggregate longer than T is cut into chunks walked by unrelated lanes; a chunk learns its incoming state
    # only afterwards (stitch kernel).  Hand-built event sequences that make the two presence hypotheses
    # differ as late and as often as possible: REQUIRE-only stretches (dropped on None, applied on Some), the first
    # MATERIALIZE / CREATE / DELETE deep inside a chunk, tombstones followed by REQUIRE runs, throws before and after
    # the deciding event, aggregates that stay None through whole chunks — with prior snapshots of every kind.
    monkeypatch.setenv("SURGE_REPLAY_CHUNK_T", str(chunk_t))
    rng = np.random.default_rng(chunk_t)
    REQ, MAT, CRE, DEL, THR, NOP = S.EVT_SET_BALANCE, S.EVT_INC, S.EVT_CREATE, S.EVT_DELETE, S.EVT_THROW, S.EVT_NOOP
    rows = []
    for length in (1, 15, 17, chunk_t, chunk_t + 1, 3 * chunk_t + 5, 8 * chunk_t, 64 * chunk_t + 9, 70 * chunk_t):
        length = min(length, 30000)
        for pattern in range(10):
            ty = np.full(length, REQ, dtype=np.int64)                   # default: REQUIRE-class everywhere
            if pattern == 1:
                ty[:] = MAT
            elif pattern == 2:
                ty[rng.integers(0, length)] = MAT                        # one deciding event somewhere
            elif pattern == 3:
                ty[rng.integers(0, length)] = CRE
            elif pattern == 4:
                ty[rng.integers(0, length)] = DEL
            elif pattern == 5:
                ty[rng.integers(0, length)] = THR                        # throw inside a REQUIRE-only aggregate
            elif pattern == 6:
                k = rng.integers(0, length, size=max(1, length // 40))
                ty[k] = rng.choice([MAT, CRE, DEL, NOP, MAT], size=k.shape[0])
            elif pattern == 7:
                ty[:] = rng.choice([REQ, REQ, REQ, MAT, DEL, NOP, S.EVT_DEC], size=length)
                ty[rng.integers(0, length)] = THR
            elif pattern == 8:
                ty[:] = rng.choice([REQ, DEL, CRE, 13], size=length, p=[0.9, 0.05, 0.049, 0.001])  # 13: unknown type = MatchError
            elif pattern == 9:
                ty[: length // 2] = DEL                                   # None through whole chunks, then REQUIRE only
            rows.append(ty)
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    lens = np.array([r.shape[0] for r in rows], dtype=np.int64)
    so = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=so[1:])
    n = int(so[-1])
    ev = np.zeros(n, dtype=S.EVENT_DTYPE)
    ev["type"] = np.concatenate(rows)
    ev["seq"] = rng.integers(1, 1 << 30, size=n)
    raw = rng.integers(-(1 << 31), 1 << 31, size=n).astype(np.int64)
    f64 = (ev["type"] == REQ) | (ev["type"] == CRE)
    vals = (rng.integers(-(1 << 20), 1 << 20, size=n) / 128.0).astype(np.float64).view(np.int64)
    ev["raw"] = np.where(f64, vals, raw & 0xFFFFFFFF).astype(np.uint64) if "raw" in ev.dtype.names else 0
    for prior_kind in ("none", "mixed"):
        prior = None
        if prior_kind == "mixed":  # Some / None / poisoned priors
            prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 4, size=lens.shape[0]), 77, synth.STRESS_MIX))
        exp = oracle.fold_csr(so, ev, prior)
        got, st = gpu_fold(so, ev, prior, algo=S.ALGO_CHUNKED)
        assert st.last_algo == S.ALGO_CHUNKED
        assert_same(got, exp, so)
        # K2t: the same chunks walked from the tile-major copy of the log (rows copied to tile boundaries, PAD events
        # behind short rows), 8- and 16-event steps
        for subs in ("1", "2"):
            monkeypatch.setenv("SURGE_REPLAY_TILED_SUBS", subs)
            got, st = gpu_fold(so, ev, prior, algo=S.ALGO_TILED)
            assert st.last_algo == S.ALGO_TILED
            assert_same(got, exp, so)
        got, _ = gpu_fold(so, ev, prior, algo=S.ALGO_FLAT)  # the flat kernel resolves presence its own way: same bytes
        assert_same(got, exp, so)


def test_tile_major_layout_prepare_reuse_and_rebind(monkeypatch):
    # K2t: the handle copies the bound log once into tile-major order (surge_replay_prepare or the first TILED fold),
    # later folds reuse the copy; a new bind drops it.  Covers: rows cut into chunks and not, empty segments, a prior
    # snapshot, a last group of fewer than 64 rows, an odd number of 8-event subtiles, redirected output, repeated folds
    # (the dispenser re-arms itself) and what surge_replay_layout_info reports.
    monkeypatch.setenv("SURGE_REPLAY_CHUNK_T", "512")
    rng = np.random.default_rng(5)
    lens = (synth.zipf_lengths(np.arange(4000, dtype=np.int64), 11) * (rng.random(4000) < 0.93)).astype(np.int64)
    lens[17] = 40_001  # cut into 79 chunks
    so, ev = synth.csr_log(lens, 12, synth.STRESS_MIX)
    prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 4, size=4000), 13, synth.STRESS_MIX))
    exp = oracle.fold_csr(so, ev, prior)
    n_events = int(so[-1])
    with ReplayEngine() as eng:
        eng.load_csr(so, ev, prior)
        assert eng.layout_info().algo == 0
        eng.prepare(S.ALGO_TILED)
        info = eng.layout_info()
        assert info.algo == S.ALGO_TILED and info.chunk_events == 512
        assert info.cut_aggregates == int((lens > 512).sum())
        assert info.virtual_rows == int(np.where(lens > 0, -(-lens // 512), 0).sum())
        assert info.tiled_bytes % 8192 == 0 and info.tiled_bytes == 16 * (n_events + info.padding_events)
        # rows are rounded up to 8 events and to their group's longest: a few percent here, < 1 % on the big logs
        assert 0 <= info.padding_events < 0.25 * n_events
        assert info.index_build_ms > 0 and info.relayout_ms > 0
        for _ in range(3):
            eng.fold(S.ALGO_TILED)
            assert_same(eng.snapshot(), exp, so)
        again = eng.layout_info()
        assert (again.relayout_ms, again.tiled_bytes) == (info.relayout_ms, info.tiled_bytes)  # built once
        eng.fold(S.ALGO_CHUNKED)  # the CSR kernels still work beside the copy
        assert_same(eng.snapshot(), exp, so)
        assert eng.layout_info().algo == S.ALGO_CHUNKED
        eng.fold(S.ALGO_TILED)
        assert_same(eng.snapshot(), exp, so)
        # a different log on the same handle: the copy is rebuilt for it
        so2, ev2 = synth.csr_log(rng.integers(0, 700, size=1000).astype(np.int64), 14, synth.STRESS_MIX)
        eng.load_csr(so2, ev2)
        assert eng.layout_info().algo == 0
        eng.fold(S.ALGO_TILED)
        assert_same(eng.snapshot(), oracle.fold_csr(so2, ev2), so2)
    # uniform fan-in (config C2's shape) through the tile-major path, both step widths
    so, ev = synth.fixed_log(5000, 256, seed=15)
    exp = oracle.fold_csr(so, ev)
    for subs in ("1", "2"):
        monkeypatch.setenv("SURGE_REPLAY_TILED_SUBS", subs)
        got, st = gpu_fold(so, ev, algo=S.ALGO_TILED)
        assert st.last_algo == S.ALGO_TILED
        assert_same(got, exp, so)


def test_auto_picks_rows_for_large_uniform_logs_and_all_uniform_kernels_agree():
    so, ev = synth.fixed_log(140_000, 32, seed=21, mix=synth.STRESS_MIX)
    exp = oracle.fold_csr(so, ev)
    got, st = gpu_fold(so, ev)
    assert st.last_algo == S.ALGO_ROWS
    assert_same(got, exp, so)
    for algo in (S.ALGO_FIXED, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED):
        got, st = gpu_fold(so, ev, algo=algo)
        assert st.last_algo == algo
        assert_same(got, exp, so)
    # rows kernel with a prior snapshot and a ragged last group (n_agg % 64 != 0)
    so, ev = synth.fixed_log(140_003, 16, seed=22, mix=synth.STRESS_MIX)
    prior = oracle.fold_csr(*synth.fixed_log(140_003, 2, seed=23, mix=synth.STRESS_MIX))
    got, st = gpu_fold(so, ev, prior, algo=S.ALGO_ROWS)
    assert_same(got, oracle.fold_csr(so, ev, prior), so)


def test_ragged_with_empty_segments_and_prior_snapshot():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 3, size=3000) * rng.integers(0, 2000, size=3000)
    so, ev = synth.csr_log(lens, 8, synth.STRESS_MIX)
    prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 4, size=3000), 9, synth.STRESS_MIX))
    got, _ = gpu_fold(so, ev, prior)
    assert_same(got, oracle.fold_csr(so, ev, prior), so)
    got, _ = gpu_fold(so, ev, prior, algo=S.ALGO_SORTED)
    assert_same(got, oracle.fold_csr(so, ev, prior), so)
    # empties at both ends and in runs
    lens = np.concatenate([np.zeros(70, np.int64), rng.integers(1, 50, 500), np.zeros(200, np.int64),
                           rng.integers(1, 3000, 30), np.zeros(65, np.int64)])
    so, ev = synth.csr_log(lens, 10, synth.STRESS_MIX)
    got, _ = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)


@pytest.mark.parametrize("lead,tail", [(1, 0), (37, 11), (1024, 5000)])
def test_csr_window_into_a_larger_events_buffer(lead, tail):
    # seg_off need not start at 0 or end at n_events (a shard's window into a bigger buffer): events outside
    # [seg_off[0], seg_off[n]) are never read into a state, whatever their (misaligned) position
    rng = np.random.default_rng(lead)
    lens = rng.integers(0, 4, size=5000) * rng.integers(0, 300, size=5000)
    so, ev = synth.csr_log(lens, 31, synth.STRESS_MIX)
    junk = synth.csr_log(np.array([lead + tail]), 32, synth.STRESS_MIX)[1]
    ev2 = np.concatenate([junk[:lead], ev, junk[lead:]])
    so2 = so + lead
    exp = oracle.fold_csr(so, ev)
    assert oracle.fold_csr(so2, ev2).tobytes() == exp.tobytes()
    for algo in (S.ALGO_AUTO, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED):
        got, _ = gpu_fold(so2, ev2, algo=algo)
        assert_same(got, exp, so)
    # a uniform log behind a lead cannot take the FIXED / ROWS fast paths, AUTO must still be right
    so, ev = synth.fixed_log(3000, 32, seed=33, mix=synth.STRESS_MIX)
    ev2 = np.concatenate([junk[:lead], ev, junk[lead:]])
    got, st = gpu_fold(so + lead, ev2)
    assert st.last_algo in (S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED)
    assert_same(got, oracle.fold_csr(so, ev), so)


def test_empty_log_and_all_empty_segments():
    so = np.zeros(1, dtype=np.int64)
    ev = S.make_events([], [], [])
    got, _ = gpu_fold(so, ev)
    assert got.shape[0] == 0
    so = np.zeros(11, dtype=np.int64)
    prior = oracle.fold_csr(*synth.fixed_log(10, 3, 1))
    got, _ = gpu_fold(so, ev, prior)
    assert got.tobytes() == prior.tobytes()
    got, _ = gpu_fold(so, ev)
    assert got.tobytes() == S.empty_states(10).tobytes()


def test_single_aggregate_longer_than_many_tasks():
    so, ev = synth.fixed_log(1, 300_000 // 16 * 16, seed=12, mix=synth.C2_MIX)
    got, _ = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)
    lens = np.array([5, 70_001, 3, 0, 2_000_003, 1], dtype=np.int64)
    so, ev = synth.csr_log(lens, 13)
    got, _ = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)


def test_segment_boundaries_on_every_tile_alignment():
    # heads at tile starts/ends, lane-chunk starts/ends, and one before/after them
    for shift in (0, 1, 15, 16, 17, 1023, 1024, 1025):
        lens = np.array([shift or 1, 1024, 16, 1, 1023 - 16, 2048, 15, 17, 1], dtype=np.int64)
        so, ev = synth.csr_log(lens, 20 + shift, synth.STRESS_MIX)
        got, _ = gpu_fold(so, ev)
        assert_same(got, oracle.fold_csr(so, ev), so)


@pytest.mark.parametrize("target_tasks,le", [(3, 16), (7, 16), (40, 16), (7, 8), (1000, 8)])
def test_flat_tasks_of_several_tiles_fetch_only_their_own_pieces_of_the_last_tile(target_tasks, le, monkeypatch):
    """A FLAT / FIXED task's last tile is fetched only up to the task's end (whole 1 KiB pieces; fold_device.h,
    issue_tile_loads): tasks of 1 ... 60 tiles whose ends fall on every kind of offset inside a tile — with the events of the
    NEXT task right behind them in memory — must still give the oracle's bytes, on None and on a prior snapshot."""
    monkeypatch.setenv("SURGE_REPLAY_TARGET_TASKS", str(target_tasks))
    monkeypatch.setenv("SURGE_REPLAY_LE_FLAT", str(le))
    monkeypatch.setenv("SURGE_REPLAY_LE_FIXED", str(le))
    rng = np.random.default_rng(77 + target_tasks)
    for lens in (rng.integers(0, 900, size=400), rng.integers(1, 40, size=9000), np.array([1, 63, 64, 65, 1000, 1023, 1024, 1025, 3000, 7, 20000, 1] * 6),
                 np.where(rng.random(3000) < 0.5, 1, rng.integers(1, 200, size=3000))):
        so, ev = synth.csr_log(lens.astype(np.int64), int(rng.integers(1, 1 << 30)), synth.STRESS_MIX)
        prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 4, size=lens.shape[0]), 5, synth.STRESS_MIX))
        for init in (None, prior):
            got, st = gpu_fold(so, ev, init, algo=S.ALGO_FLAT)
            assert st.n_tasks >= min(target_tasks, 2) - 1
            assert_same(got, oracle.fold_csr(so, ev, init), so)
    so, ev = synth.fixed_log(517, 48, seed=9, mix=synth.STRESS_MIX)  # FIXED: 24 816 events, tasks end inside a tile
    got, _ = gpu_fold(so, ev, algo=S.ALGO_FIXED)
    assert_same(got, oracle.fold_csr(so, ev), so)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SURGE_TEST_FUZZ_SEEDS", "6"))))
def test_random_log_shapes_through_every_kernel(seed):
    # shape fuzz: segment-length distributions that stress different paths (all short, a few giants, runs of
    # empties, lengths around the 8/16/64-event tile edges), random type mixes, with and without a prior snapshot;
    # every kernel that accepts the shape must give the oracle's bytes
    rng = np.random.default_rng(1000 + seed)
    for _ in range(12):
        n = int(rng.integers(1, 3000))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            lens = rng.integers(0, 6, size=n)
        elif kind == 1:
            lens = rng.integers(0, 40, size=n)
            lens[rng.integers(0, n, size=max(1, n // 200))] = rng.integers(2000, 20000)
        elif kind == 2:
            lens = rng.choice([0, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025], size=n)
        elif kind == 3:
            lens = np.where(rng.random(n) < 0.7, 0, rng.integers(1, 300, size=n))
        else:
            lens = np.full(n, int(rng.choice([16, 32, 48, 256])))
        mix = [synth.C1_MIX, synth.C2_MIX, synth.STRESS_MIX][int(rng.integers(0, 3))]
        so, ev = synth.csr_log(lens.astype(np.int64), int(rng.integers(1, 1 << 30)), mix)
        prior = None
        if rng.random() < 0.5:
            prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 4, size=n), int(rng.integers(1, 1 << 30)), synth.STRESS_MIX))
        exp = oracle.fold_csr(so, ev, prior)
        algos = [S.ALGO_AUTO, S.ALGO_FLAT, S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED, S.ALGO_SHORT] + ([S.ALGO_FIXED, S.ALGO_ROWS] if kind == 4 else [])
        for algo in algos:
            got, _ = gpu_fold(so, ev, prior, algo=algo)
            assert got.tobytes() == exp.tobytes(), (seed, kind, n, algo)


def test_every_aggregate_poisoned_or_deleted():
    n = 3000
    ev = S.make_events(np.tile([S.EVT_INC, S.EVT_THROW, S.EVT_INC], n), np.tile([1, 2, 3], n), np.tile([4, 0, 9], n))
    so = np.arange(n + 1, dtype=np.int64) * 3
    got, st = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)
    assert st.n_poisoned == n and (got["count"] == 4).all()
    ev["type"][1::3] = S.EVT_DELETE
    got, st = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)
    assert st.n_poisoned == 0 and (got["count"] == 9).all() and (got["event_count"] == 1).all()


def test_int32_and_int64_wrap_match_the_jvm():
    n = 40000
    ev = S.make_events(np.full(n, S.EVT_INC), np.arange(1, n + 1), np.full(n, S.INT32_MAX))
    so = np.array([0, n], dtype=np.int64)
    got, _ = gpu_fold(so, ev)
    assert_same(got, oracle.fold_csr(so, ev), so)
    assert got[0]["count"] == np.int64(n * S.INT32_MAX).astype(np.int32)  # wrapped
    assert got[0]["sum64"] == n * S.INT32_MAX


def test_custom_algebras_counter_and_bank_account():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 300, size=2000)
    so = np.zeros(lens.size + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    n = int(so[-1])
    ev = S.make_events(rng.integers(0, 4, n), rng.integers(0, 1 << 30, n), rng.integers(-(1 << 31), 1 << 31, n))
    ev["type"][rng.random(n) < 0.97] %= 3  # few throwing events
    got, _ = gpu_fold(so, ev, algebra=COUNTER_ALGEBRA)
    assert_same(got, oracle.fold_csr(so, ev, None, COUNTER_ALGEBRA), so)
    ev = S.make_events(rng.integers(0, 2, n), np.zeros(n), values=rng.standard_normal(n))
    got, _ = gpu_fold(so, ev, algebra=BANK_ACCOUNT_ALGEBRA)
    assert_same(got, oracle.fold_csr(so, ev, None, BANK_ACCOUNT_ALGEBRA), so)


def test_count_set_op_and_nan_payload_bits():
    alg = S.EventAlgebra(desc=(S.CLS_MATERIALIZE | S.D_COUNT_SET | S.D_EVCOUNT_INC, S.CLS_MATERIALIZE | S.D_COUNT_ADD,
                               S.CLS_REQUIRE | S.D_BALANCE_SET, S.CLS_CREATE))
    rng = np.random.default_rng(5)
    n = 50000
    so = np.arange(0, n + 1, 50, dtype=np.int64)
    ev = S.make_events(rng.integers(0, 4, n), np.arange(n), rng.integers(-100, 100, n))
    nan_bits = np.uint64(0x7FF8DEADBEEF0001)  # a NaN payload must be moved bit-exactly, never canonicalised
    ev["raw"][ev["type"] == 2] = nan_bits
    got, _ = gpu_fold(so, ev, algebra=alg)
    exp = oracle.fold_csr(so, ev, None, alg)
    assert_same(got, exp, so)
    assert (got.view(np.uint8).reshape(-1, 64)[:, 16:24].view(np.uint64) == nan_bits).any()


def test_micro_batch_append_fold_matches_full_refold():
    # K3: fold(prev snapshot, new events) == fold over the concatenated log (associativity of the monoid)
    rng = np.random.default_rng(7)
    n_agg = 5000
    so, ev = synth.zipf_log(n_agg, 30, max_len=256, mix=synth.STRESS_MIX)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        state = oracle.fold_csr(so, ev)
        for batch in range(4):
            m = 20000
            agg_idx = rng.integers(0, n_agg, m)
            be = synth.to_event_records(synth.event_words(np.arange(m, dtype=np.int64) + batch * m, agg_idx,
                                                           np.arange(m, dtype=np.int64), 31, synth.STRESS_MIX))
            from surge_amd.log import batch_groups

            group_agg, group_off, sorted_ev = batch_groups(agg_idx, be)
            if batch % 2 == 0:
                eng.append_fold(group_agg, group_off, sorted_ev)  # caller groups
            else:
                eng.append_events(agg_idx, be)  # library groups (stable radix sort), same result
            # oracle: fold each group onto the previous state
            full_off = np.zeros(n_agg + 1, np.int64)
            np.cumsum(np.bincount(agg_idx, minlength=n_agg), out=full_off[1:])
            state = oracle.fold_csr(full_off, sorted_ev, state)
            got = eng.snapshot()
            assert_same(got, state, full_off)


@pytest.mark.parametrize("seed", range(4))
def test_micro_batch_fuzz_sizes_skews_and_group_counts(seed):
    # K3 over many batch shapes: one event, one hot aggregate taking the whole batch, every aggregate once, batches
    # larger than a wave task, aggregates far beyond 2^16 (several radix passes in the library's group-by)
    from surge_amd.log import batch_groups

    rng = np.random.default_rng(500 + seed)
    n_agg = int(rng.choice([1, 70, 5000, 200_000]))
    so, ev = synth.csr_log(rng.integers(0, 5, size=n_agg), 61 + seed, synth.STRESS_MIX)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        state = oracle.fold_csr(so, ev)
        base = 0
        for shape in ("one", "hot", "each", "big", "zipf", "one"):
            if shape == "one":
                agg_idx = rng.integers(0, n_agg, 1)
            elif shape == "hot":
                agg_idx = np.full(3000, int(rng.integers(0, n_agg)))
            elif shape == "each":
                agg_idx = rng.permutation(min(n_agg, 50_000))
            elif shape == "big":
                agg_idx = rng.integers(0, n_agg, 70_000)
            else:
                agg_idx = np.minimum((rng.pareto(1.2, 20_000) * 3).astype(np.int64), n_agg - 1)
            m = agg_idx.shape[0]
            be = synth.to_event_records(synth.event_words(np.arange(m, dtype=np.int64) + base, agg_idx.astype(np.int64),
                                                           np.arange(m, dtype=np.int64), 71 + seed, synth.STRESS_MIX))
            base += m
            eng.append_events(agg_idx.astype(np.int64), be)
            _, _, sorted_ev = batch_groups(agg_idx.astype(np.int64), be)
            full_off = np.zeros(n_agg + 1, np.int64)
            np.cumsum(np.bincount(agg_idx, minlength=n_agg), out=full_off[1:])
            state = oracle.fold_csr(full_off, sorted_ev, state)
            assert_same(eng.snapshot(), state, full_off)


def test_micro_batches_pipeline_without_host_syncs_and_bad_device_batches_are_skipped():
    # K3 never waits for the device (v1): 40 host batches of changing sizes go in back to back through the two pinned
    # staging areas, the group count stays on the device (plan_dev_kernel), one snapshot at the end must equal the oracle's
    # batch-by-batch replay.  A device-resident batch with an index out of range is skipped as a whole ON THE DEVICE and
    # reported once by the next synchronize; a host batch with a bad index is refused before anything is enqueued.
    import torch

    from surge_amd.log import batch_groups

    rng = np.random.default_rng(11)
    n_agg = 30_000
    so, ev = synth.csr_log(rng.integers(0, 6, size=n_agg), 91, synth.STRESS_MIX)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        state = oracle.fold_csr(so, ev)
        base = 0
        for b in range(40):
            m = int(rng.choice([1, 37, 5_000, 60_000, 130_000]))
            agg_idx = rng.integers(0, n_agg, m).astype(np.int64)
            be = synth.to_event_records(synth.event_words(np.arange(m, dtype=np.int64) + base, agg_idx, np.arange(m, dtype=np.int64), 92, synth.STRESS_MIX))
            base += m
            eng.append_events(agg_idx, be)  # no synchronize, no snapshot in between
            group_agg, group_off, sorted_ev = batch_groups(agg_idx, be)
            state[group_agg] = oracle.fold_csr(group_off, sorted_ev, state[group_agg])
        assert eng.snapshot().tobytes() == state.tobytes()
        # host batch with a bad index: immediate, nothing applied
        bad = np.array([5, n_agg, 7], dtype=np.int64)
        be3 = synth.to_event_records(synth.event_words(np.arange(3, dtype=np.int64), bad, np.arange(3, dtype=np.int64), 93, synth.STRESS_MIX))
        with pytest.raises(ReplayError) as ei:
            eng.append_events(bad, be3)
        assert ei.value.status == -6
        # device batches: good, BAD (skipped on the device), good — all enqueued before the host looks
        dev = torch.device("cuda:0")
        batches = []
        for k in range(3):
            m = 9_000
            agg_idx = rng.integers(0, n_agg, m).astype(np.int64)
            if k == 1:
                agg_idx[4321] = n_agg + 17
            words = synth.event_words(np.arange(m, dtype=np.int64) + base, np.minimum(agg_idx, n_agg - 1), np.arange(m, dtype=np.int64), 94, synth.STRESS_MIX)
            base += m
            batches.append((agg_idx, words))
            eng.append_events(torch.from_numpy(agg_idx).to(dev), torch.from_numpy(words).to(dev))
        with pytest.raises(ReplayError) as ei:
            eng.synchronize()
        assert ei.value.status == -6 and "1 micro-batch" in str(ei.value)
        eng.synchronize()  # reported once
        for k in (0, 2):
            agg_idx, words = batches[k]
            group_agg, group_off, sorted_ev = batch_groups(agg_idx, synth.to_event_records(words))
            state[group_agg] = oracle.fold_csr(group_off, sorted_ev, state[group_agg])
        assert eng.snapshot().tobytes() == state.tobytes()


def test_get_point_reads_and_errors():
    so, ev = synth.fixed_log(100, 32, 4)
    with ReplayEngine() as eng:
        with pytest.raises(ReplayError) as ei:
            eng.fold()
        assert ei.value.status == -2  # SURGE_E_STATE
        eng.load_csr(so, ev)
        eng.fold()
        exp = oracle.fold_csr(so, ev)
        for a in (0, 17, 99):  # served by a device read (no snapshot yet)
            assert eng.get_raw(a).tobytes() == exp[a].tobytes()
        eng.snapshot()
        for a in (0, 17, 99):  # served by the host mirror
            assert eng.get_raw(a).tobytes() == exp[a].tobytes()
        with pytest.raises(ReplayError) as ei:
            eng.get_raw(100)
        assert ei.value.status == -6  # SURGE_E_RANGE
        bad = so.copy()
        bad[5] = bad[6] + 1
        with pytest.raises(ReplayError) as ei:
            eng.load_csr(bad, ev)
        assert ei.value.status == -1 and "monotone" in str(ei.value)
        with pytest.raises(ReplayError):
            eng.fold(99)  # not an algorithm
    so, ev = synth.fixed_log(10, 20, 4)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        with pytest.raises(ReplayError) as ei:
            eng.fold(S.ALGO_FIXED)  # 20 % 16 != 0
        assert ei.value.status == -5


def test_device_resident_binding_and_idempotence():
    import torch

    dev = torch.device("cuda:0")
    so, ev = synth.fixed_log_device(20000, 256, 2, dev)
    out = torch.empty((20000, 64), dtype=torch.uint8, device=dev)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev, None, out)
        eng.fold()
        eng.synchronize()
        first = out.clone()
        eng.fold()  # replay is a pure function of the log
        eng.synchronize()
        assert torch.equal(first, out)
        view = eng.device_state()
        assert view.data_ptr() == out.data_ptr()
    exp = oracle.fold_csr(so.cpu().numpy(), synth.to_event_records(ev))
    assert out.cpu().numpy().tobytes() == exp.tobytes()


def test_partition_hash_kernel_matches_cpu_entry_point_and_oracle():
    import torch

    from surge_amd.kafka import partition_for_keys, utf16_table

    keys = [f"acct-{i:08d}:{i % 13}" for i in range(20000)] + ["", ":", "a", "ab:c", "ünï-✓", "x" * 301]
    data, off = utf16_table(keys)
    with ReplayEngine() as eng:
        for cut in (False, True):  # partitionForKey of the whole string / after PartitionStringUpToColon.partitionBy
            cpu = partition_for_keys(keys, 64, up_to_colon=cut)
            d_out = torch.zeros(len(keys), dtype=torch.int32, device="cuda:0")
            eng.partition_hash_device(torch.from_numpy(data.view(np.int16)).cuda(), torch.from_numpy(off).cuda(), 64, d_out,
                                      up_to_colon=cut)
            eng.synchronize()
            assert (d_out.cpu().numpy() == cpu).all()
            assert (oracle.partition_hash_batch(data, off, 64, up_to_colon=cut) == cpu).all()
    assert (partition_for_keys(keys, 64) != partition_for_keys(keys, 64, up_to_colon=True)).any()


def test_gpu_json_encoder_matches_play_json_text_of_the_counter_fixture():
    # N3: bulk serialized state == Json.toJson(state).toString() (TestBoundedContext.scala:127-129), byte for byte
    import torch

    from surge_amd.encode import JsonTemplate, encode_states, key_table_utf8
    from fixture_models import CounterAggregateFormat, State

    n = 5000
    rng = np.random.default_rng(11)
    keys = [f"agg-{i:05d}" for i in range(n)]
    keys[7], keys[8], keys[9], keys[10] = 'we"ird\\id', "tab\there\nnl", "ünï-✓-ключ", "\x01\x1f"
    lens = rng.integers(0, 12, size=n)
    so, ev = synth.csr_log(lens, 12, synth.STRESS_MIX)
    ev["raw"][(ev["type"] == S.EVT_INC) & (rng.random(ev.shape[0]) < 0.3)] = np.uint64(np.uint32(np.int32(-7)))
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        states = eng.snapshot()
        data, off = key_table_utf8(keys)
        d_out, d_off = encode_states(eng, JsonTemplate.counter(), torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda())
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
    fmt = CounterAggregateFormat()
    n_emitted = 0
    for a in range(n):
        text = out[offs[a]:offs[a + 1]]
        fl = int(states[a]["flags"])
        if fl == S.STATE_PRESENT:
            n_emitted += 1
            assert text == fmt.write_state(State(keys[a], int(states[a]["count"]), int(states[a]["version"]))).value
            assert text == oracle.counter_state_json(keys[a], int(states[a]["count"]), int(states[a]["version"]))
        else:
            assert text == b""  # None => tombstone, poisoned => nothing
    assert 0 < n_emitted < n


@pytest.mark.gpu
def test_gpu_json_encoder_writes_bank_account_states_with_play_json_double_text():
    # R14 / N3: Json.toJson(BankAccount).toString() (BankAccountSurgeModel.scala:26-28) from the device, byte for byte:
    # the Double as play-json 2.9.2 writes it, owner / security code from side string columns
    import uuid

    import torch

    from surge_amd.encode import JsonTemplate, encode_states, key_table_utf8
    from surge_amd.replay import ReplayError
    from fixture_models import BANK_ACCOUNT_ALGEBRA, BA_CREATED, BA_UPDATED, BankAccount, BankAccountFormat

    n = 20000
    rng = np.random.default_rng(21)
    keys = [str(uuid.UUID(int=int(x))) for x in rng.integers(0, 1 << 62, size=n)]
    owners = [f"Owner {i} \"q\" ünï" if i % 97 == 0 else f"Jane Doe {i}" for i in range(n)]
    codes = ["" if i % 50 == 0 else f"{i % 10000:04d}" for i in range(n)]
    # one Created (+ for half of them an Updated) per account; balances of every magnitude, cents, integers, special values
    two = rng.random(n) < 0.5
    lens = 1 + two.astype(np.int64)
    so = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    ev = np.zeros(int(so[-1]), dtype=S.EVENT_DTYPE)
    ev["type"][so[:-1]] = BA_CREATED
    ev["type"][so[:-1][two] + 1] = BA_UPDATED
    kinds = rng.integers(0, 6, size=ev.shape[0])
    vals = np.select([kinds == 0, kinds == 1, kinds == 2, kinds == 3, kinds == 4],
                     [np.round(rng.random(ev.shape[0]) * 1e7) / 100, rng.integers(-10 ** 6, 10 ** 6, size=ev.shape[0]).astype(np.float64),
                      rng.random(ev.shape[0]) * 10.0 ** rng.integers(-12, 25, size=ev.shape[0]), rng.standard_normal(ev.shape[0]) * 1e3,
                      rng.choice([0.0, -0.0, 1e20, 1e-7, 5e-324, 1.7976931348623157e308, 0.1 + 0.2, 1e21, 100.0], size=ev.shape[0])],
                     default=rng.integers(0, 0x7FF0000000000000, size=ev.shape[0], dtype=np.uint64).view(np.float64))
    ev["raw"] = vals.view(np.uint64)
    bad = rng.choice(ev.shape[0], size=25, replace=False)
    ev["raw"][bad] = rng.choice(np.array([0x7FF8000000000000, 0x7FF0000000000000, 0xFFF0000000000000], dtype=np.uint64), size=25)
    fmt = BankAccountFormat()
    with ReplayEngine(BANK_ACCOUNT_ALGEBRA) as eng:
        eng.load_csr(so, ev)
        eng.fold()
        states = eng.snapshot()
        data, off = key_table_utf8(keys)
        cols = [tuple(torch.from_numpy(x).cuda() for x in key_table_utf8(col)) for col in (owners, codes)]
        nonfinite = ~np.isfinite(states["balance"])
        assert 0 < nonfinite.sum() <= 25
        with pytest.raises(ReplayError, match="NaN / infinite"):  # reported, after everything else was encoded ...
            encode_states(eng, JsonTemplate.bank_account(), torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda(), strings=cols)
        # ... so encode again with those aggregates filtered out (what a publisher that already knows them would do)
        kind = torch.from_numpy(np.where(nonfinite, 0, 1).astype(np.uint8)).cuda()
        eng._check(eng._lib.surge_replay_set_encode_filter(eng._h, ctypes.c_void_p(kind.data_ptr())))
        d_out, d_off = encode_states(eng, JsonTemplate.bank_account(), torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda(), strings=cols)
        eng._check(eng._lib.surge_replay_set_encode_filter(eng._h, None))
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
    for a in range(n):
        text = out[offs[a]:offs[a + 1]]
        if nonfinite[a]:
            assert text == b""
            continue
        bal = float(states[a]["balance"])
        # the oracle-side restatement (repr digits + BigDecimal rules), then the host plugin's writeState
        exp = (f'{{"accountNumber":"{keys[a]}","accountOwner":{json.dumps(owners[a], ensure_ascii=False)},'
               f'"securityCode":"{codes[a]}","balance":{oracle.play_json_double_text(bal)}}}').encode("utf-8")
        assert text == exp, (a, text, exp)
        assert text == fmt.write_state(BankAccount(uuid.UUID(keys[a]), owners[a], codes[a], bal)).value
        assert float(json.loads(text)["balance"]) == bal  # and it reads back as the same Double


@pytest.mark.gpu
@pytest.mark.parametrize("publish_mirror", [True, False])
def test_point_reads_tolerate_32_concurrent_readers(publish_mirror):
    # S2 is called from the reference's 32-thread IO pool (ThreadPools.scala:10-11): point reads must be safe under
    # that concurrency, both against a published host mirror and when every read goes to the device
    from concurrent.futures import ThreadPoolExecutor

    n = 20000
    so, ev = synth.csr_log(np.random.default_rng(9).integers(0, 20, size=n), 51, synth.STRESS_MIX)
    exp = oracle.fold_csr(so, ev)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        if publish_mirror:
            eng.snapshot()

        def reader(seed):
            rng = np.random.default_rng(seed)
            bad = 0
            for a in rng.integers(0, n, size=300 if publish_mirror else 60):
                bad += eng.get_raw(int(a)).tobytes() != exp[int(a)].tobytes()
            return bad

        with ThreadPoolExecutor(max_workers=32) as pool:
            assert sum(pool.map(reader, range(32))) == 0


def _protobuf_state_class():
    """message State { string aggregateId = 1; bytes payload = 2; } (multilanguage-protocol.proto:7-10), built with
    the real protobuf runtime so the expected bytes come from Google's encoder, not from a restatement."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    fd = descriptor_pb2.FileDescriptorProto(name="surge_multilanguage_state.proto", syntax="proto3")
    m = fd.message_type.add(name="State")
    F = descriptor_pb2.FieldDescriptorProto
    m.field.add(name="aggregateId", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    m.field.add(name="payload", number=2, type=F.TYPE_BYTES, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("State"))


@pytest.mark.gpu
def test_gpu_protobuf_state_envelope_matches_the_protobuf_runtime():
    # N3, multilanguage flavour: the stored value is protobuf State{aggregateId, payload}.toByteArray
    # (GenericSurgeCommandBusinessLogic.scala:36-39); payload = the SDK's serialized state, here the Counter JSON
    import torch

    from surge_amd.encode import JsonTemplate, encode_states, key_table_utf8

    State = _protobuf_state_class()
    n = 3000
    rng = np.random.default_rng(5)
    keys = [f"agg-{i}" for i in range(n)]
    keys[3] = ""                       # proto3 omits an empty string field
    keys[4] = "k" * 127                # 1-byte varint boundary
    keys[5] = "k" * 128                # 2-byte varint
    keys[6] = "ключ-✓" * 40            # multi-byte UTF-8, payload > 127 bytes too
    keys[7] = "q\"uote" * 30           # escaped in the JSON payload, raw in field 1
    keys[8] = "z" * 20000              # 3-byte varints; also forces the un-staged block path
    lens = rng.integers(0, 5, size=n)
    lens[3:9] = 2
    so, ev = synth.csr_log(lens, 41, synth.STRESS_MIX)
    ev["type"][so[3]:so[9]] = S.EVT_INC  # the special keys must be Some(...)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        states = eng.snapshot()
        assert all(int(states[a]["flags"]) == S.STATE_PRESENT for a in range(3, 9))
        data, off = key_table_utf8(keys)
        d_out, d_off = encode_states(eng, JsonTemplate.counter(), torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda(),
                                     envelope="protobuf_state")
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
    n_emitted = 0
    for a in range(n):
        got = out[offs[a]:offs[a + 1]]
        if int(states[a]["flags"]) == S.STATE_PRESENT:
            n_emitted += 1
            payload = oracle.counter_state_json(keys[a], int(states[a]["count"]), int(states[a]["version"]))
            assert got == State(aggregateId=keys[a], payload=payload).SerializeToString(), a
            back = State.FromString(got)
            assert back.aggregateId == keys[a] and back.payload == payload
        else:
            assert got == b""
    assert 0 < n_emitted < n and offs[n] == len(out)


@pytest.mark.gpu
def test_sdk_sample_model_replays_and_encodes_to_its_stored_protobuf_form():
    # R8 end to end: deposits folded on the GPU under the sample's one-type algebra, then encoded in bulk as
    # State{aggregateId, payload = {"balance":N}} — byte-equal to what the multilanguage gateway stores
    import torch

    from surge_amd.encode import JP_I32, JsonTemplate, encode_states, key_table_utf8
    from fixture_models import SDK_SAMPLE_MODEL, MoneyDeposited, SdkBankAccount, SdkEvent, SdkSampleCommandModel, sdk_sample_state_bytes

    rng = np.random.default_rng(77)
    model = SdkSampleCommandModel()
    n = 700
    keys = [f"0c3f1d9e-7a55-4a5c-9d5e-{i:012x}" for i in range(n)]
    per_agg = [[MoneyDeposited(int(a)) for a in rng.integers(0, 2**31, size=int(k))] for k in rng.integers(0, 40, size=n)]
    so = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(d) for d in per_agg], out=so[1:])
    ev = model.encode_events([SdkEvent(keys[a], d) for a in range(n) for d in per_agg[a]])
    with ReplayEngine(model.event_algebra()) as eng:
        eng.load_csr(so, ev)
        eng.fold()
        states = eng.snapshot()
        data, off = key_table_utf8(keys)
        tmpl = JsonTemplate((b'{"balance":', (JP_I32, 0), b"}"))
        d_out, d_off = encode_states(eng, tmpl, torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda(), envelope="protobuf_state")
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
    for a in range(n):
        want = SDK_SAMPLE_MODEL.apply_events(None, per_agg[a])
        got = out[offs[a]:offs[a + 1]]
        if want is None:
            assert got == b"" and not int(states[a]["flags"]) & S.STATE_PRESENT
        else:
            assert model.state_from_fixed(keys[a], states[a]) == want
            assert got == sdk_sample_state_bytes(keys[a], want)


@pytest.mark.gpu
@pytest.mark.parametrize("n,long_every", [(1, 0), (255, 0), (256, 0), (257, 0), (1500, 0), (1500, 3)])
def test_gpu_json_encoder_block_staging_and_long_key_fallback(n, long_every):
    # the write pass composes a block's 256 values in LDS; blocks whose text exceeds the staging buffer
    # (every third key ~600 bytes of escapes here) write straight to global.  Both must give the same bytes.
    import torch

    from surge_amd.encode import JsonTemplate, encode_states, key_table_utf8

    rng = np.random.default_rng(n * 7 + long_every)
    keys = [f"k{i}" * (1 + i % 5) for i in range(n)]
    if long_every:
        for i in range(0, n, long_every):
            keys[i] = ("\x02long\"" * 40) + str(i)
    lens = rng.integers(0, 6, size=n)
    so, ev = synth.csr_log(lens, 21, synth.STRESS_MIX)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        states = eng.snapshot()
        data, off = key_table_utf8(keys)
        d_out, d_off = encode_states(eng, JsonTemplate.counter(), torch.from_numpy(data).cuda(), torch.from_numpy(off).cuda())
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
    want = b"".join(oracle.counter_state_json(keys[a], int(states[a]["count"]), int(states[a]["version"]))
                    for a in range(n) if int(states[a]["flags"]) == S.STATE_PRESENT)
    assert out[: offs[n]] == want and offs[n] == len(want)


def test_append_fold_rejects_a_batch_that_names_an_aggregate_twice():
    so, ev = synth.fixed_log(10, 16, 4)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        with pytest.raises(ReplayError) as ei:
            eng.append_fold(np.array([3, 3]), np.array([0, 1, 2]), ev[:2])
        assert ei.value.status == -1 and "more than one group" in str(ei.value)


def test_two_engines_on_two_streams_do_not_interfere():
    import torch

    dev = torch.device("cuda:0")
    logs = [synth.fixed_log_device(150_000, 32, 40 + i, dev) for i in range(2)]
    outs = [torch.empty((150_000, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    engines = [ReplayEngine() for _ in range(2)]
    try:
        torch.cuda.synchronize()
        for e, s, (so, ev), o in zip(engines, streams, logs, outs):
            e.use_stream(s)
            e.load_csr(so, ev, None, o)
        for _ in range(5):  # interleaved launches on both streams
            for e in engines:
                e.fold()
        for e in engines:
            e.synchronize()
        for (so, ev), o in zip(logs, outs):
            exp = oracle.fold_csr(so[:3001].cpu().numpy(), synth.to_event_records(ev[: 3000 * 32]))
            assert o[:3000].cpu().numpy().tobytes() == exp.tobytes()
    finally:
        for e in engines:
            e.close()


def test_snapshot_wire_form_round_trips():
    import torch

    so, ev = synth.zipf_log(3000, 17, max_len=64, mix=synth.STRESS_MIX)
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        eng.fold()
        eng.synchronize()
        st = eng.device_state()
        packed = torch.empty((3000, 40), dtype=torch.uint8, device="cuda:0")
        back = torch.full((3000, 64), 0xAB, dtype=torch.uint8, device="cuda:0")
        side = torch.cuda.Stream()
        eng.pack_states(st, packed, stream=side)
        eng.unpack_states(packed, back, stream=side)
        side.synchronize()
        assert torch.equal(packed, st[:, :40])
        assert torch.equal(back, st)  # the reserved tail is always zero, so 40 bytes carry the whole state


def test_snapshot_gather_on_the_rccl_backend_single_rank():
    """The CUDA code path of SnapshotGather (side stream, pack -> exchange -> unpack, events) through the
    real RCCL backend with one rank (a 1-GPU box cannot host more; the multi-rank logic is covered by the
    gloo tests)."""
    import socket

    import torch
    import torch.distributed as dist

    from surge_amd.dist import SnapshotGather

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        so, ev = synth.fixed_log_device(70_000, 32, 8, "cuda:0")
        with ReplayEngine() as eng:
            compute = torch.cuda.Stream()
            eng.use_stream(compute)
            for mode, packed in (("p2p", True), ("allgather", True), ("allgather", False)):
                g = SnapshotGather(70_000, "cuda:0", mode=mode, engine=eng, packed=packed)
                bufs = g.make_local_buffers()
                torch.cuda.synchronize()
                eng.load_csr(so, ev, None, bufs[0])
                done = torch.cuda.Event()
                for step in range(3):
                    slot = step & 1
                    g.wait(slot, compute)
                    eng.set_state_out(bufs[slot])
                    eng.fold()
                    done.record(compute)
                    g.launch(slot, bufs[slot], done)
                torch.cuda.synchronize()
                for slot in (0, 1):
                    assert torch.equal(g.result(slot)[0], bufs[slot])
            exp = oracle.fold_csr(so[:2001].cpu().numpy(), synth.to_event_records(ev[: 2000 * 32]))
            assert bufs[0][:2000].cpu().numpy().tobytes() == exp.tobytes()
    finally:
        dist.destroy_process_group()


def _random_algebra(rng):
    """1 .. 16 event types with random classes and field ops, a few of them throwing, random defaults"""
    n_types = int(rng.integers(1, S.MAX_EVENT_TYPES + 1))
    desc = []
    for _ in range(n_types):
        d = int(rng.choice([S.CLS_MATERIALIZE, S.CLS_REQUIRE, S.CLS_CREATE, S.CLS_DELETE], p=[0.45, 0.35, 0.12, 0.08]))
        d |= int(rng.choice([0, S.D_COUNT_ADD, S.D_COUNT_SUB, S.D_COUNT_SET])) | int(rng.choice([0, S.D_SUM_ADD, S.D_SUM_SUB]))
        for bit in (S.D_VERSION_SET, S.D_BALANCE_SET, S.D_MIN_ARG, S.D_MAX_ARG, S.D_EVCOUNT_INC):
            d |= bit if rng.random() < 0.4 else 0
        if rng.random() < 0.06:
            d |= S.D_POISON
        desc.append(d)
    if rng.random() < 0.4:  # a narrow schema: one or two fields in use, so whole parts of the walk fold away in the compiled build
        keep = S.CLS_MASK | S.D_POISON | int(rng.choice([S.D_COUNT_MASK | S.D_VERSION_SET, S.D_SUM_MASK, S.D_BALANCE_SET | S.D_EVCOUNT_INC, S.D_MIN_ARG | S.D_MAX_ARG]))
        desc = [d & keep for d in desc]
    return S.EventAlgebra(desc=tuple(desc), default_count=int(rng.integers(-5, 5)), default_version=int(rng.integers(0, 3)),
                          default_sum64=int(rng.integers(-1 << 40, 1 << 40)), default_balance=float(rng.standard_normal()),
                          default_event_count=int(rng.integers(0, 3)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("SURGE_TEST_FUZZ_SEEDS", "6"))))
def test_the_flat_kernel_compiled_for_a_v1_schema_equals_the_interpreter_and_the_oracle(seed, monkeypatch):
    """a v1 handle compiles the flat kernel for its op table at its first flat fold (hiprtc: the table's
    words become compile-time masks over the event type, no LDS copy of the table; SURGE_REPLAY_RTC=0 keeps the
    ahead-of-time build that reads the table from LDS).  Random schemas — 1 .. 16 types, every class and field op, throwing
    types, narrow ones whose unused fields fold away — on random log shapes with event types beyond the schema (poison),
    with and without a prior snapshot: whole logs (ALGO_FLAT) and micro-batches onto the resident state (K3), both builds,
    bit for bit the oracle's states."""
    rng = np.random.default_rng(4000 + seed)
    for _ in range(3):
        alg = _random_algebra(rng)
        n = int(rng.integers(1, 2500))
        lens = [rng.integers(0, 6, size=n), rng.integers(0, 60, size=n), np.where(rng.random(n) < 0.98, rng.integers(0, 9, size=n), rng.integers(500, 9000, size=n))][int(rng.integers(0, 3))]
        so = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=so[1:])
        m = int(so[-1])
        ty = rng.integers(0, len(alg.desc) + (2 if rng.random() < 0.3 else 0), m)  # now and then a type the schema does not know
        ev = S.make_events(ty, rng.integers(0, 1 << 30, m), rng.integers(-(1 << 31), 1 << 31, m))
        prior = None
        if rng.random() < 0.5:
            pl = rng.integers(0, 4, size=n)
            po = np.zeros(n + 1, np.int64)
            np.cumsum(pl, out=po[1:])
            pe = S.make_events(rng.integers(0, len(alg.desc), int(po[-1])), rng.integers(0, 1 << 30, int(po[-1])), rng.integers(-1000, 1000, int(po[-1])))
            prior = oracle.fold_csr(po, pe, None, alg)
            if rng.random() < 0.6:
                # a snapshot written under another model version: present aggregates hold arbitrary values in EVERY field, also
                # in the ones this schema never touches (those must come out as they went in unless the aggregate is re-created)
                live = (prior["flags"] & S.STATE_PRESENT) != 0
                for f in ("count", "version", "min_arg", "max_arg"):
                    prior[f][live] = rng.integers(-1 << 31, 1 << 31, int(live.sum()))
                prior["sum64"][live] = rng.integers(-1 << 62, 1 << 62, int(live.sum()))
                prior["event_count"][live] = rng.integers(0, 1 << 32, int(live.sum()))
                prior["balance"][live] = rng.standard_normal(int(live.sum()))
        exp = oracle.fold_csr(so, ev, prior, alg)
        # micro-batches: the same events in a random interleaving of the aggregates (each aggregate's own events in order)
        topic_agg = rng.permutation(np.repeat(np.arange(n, dtype=np.int64), lens))
        src = np.zeros(m, np.int64)
        src[np.argsort(topic_agg, kind="stable")] = np.arange(m)  # the k-th record of aggregate a is its k-th event
        for build in ("1", "0"):
            monkeypatch.setenv("SURGE_REPLAY_RTC", build)
            with ReplayEngine(alg) as eng:
                info = eng.kernel_info()
                assert info["specialised"] == (build == "1"), info
                eng.load_csr(so, ev, prior)
                eng.fold(S.ALGO_FLAT)
                assert eng.snapshot().tobytes() == exp.tobytes(), (seed, build, alg.desc)
                if m:
                    eng.load_csr(np.zeros(n + 1, np.int64), np.zeros(0, dtype=S.EVENT_DTYPE), prior)
                    eng.fold()
                    for c0 in range(0, m, m // 3 + 1):
                        eng.append_events(topic_agg[c0:c0 + m // 3 + 1], ev[src[c0:c0 + m // 3 + 1]])
                    eng.synchronize()
                    assert eng.snapshot().tobytes() == exp.tobytes(), (seed, build, "K3", alg.desc)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SURGE_TEST_FUZZ_SEEDS", "6"))))
def test_the_lane_kernels_compiled_for_a_v1_schema_equal_the_interpreter_and_the_oracle(seed, monkeypatch):
    """SORTED / CHUNKED / ROWS run kernels compiled for the handle's op table (hiprtc, the V1_LANES program:
    table words are comparisons of the event type with compile-time constants, fields no event type touches leave the walk;
    SURGE_REPLAY_RTC_LANES=0 keeps the ahead-of-time kernels that read the table from LDS).  Random schemas — every class
    and field op, throwing types, narrow ones — on ragged logs (whole aggregates and, with a small chunk target, chunks of cut
    aggregates whose presence is resolved by the stitch kernel) and uniform logs, 8- and 16-event lanes, event types beyond
    the schema, priors written under other schemas: both builds, bit for bit the oracle's states."""
    rng = np.random.default_rng(6000 + seed)
    for it in range(3):
        alg = _random_algebra(rng)
        n = int(rng.integers(1, 2500))
        uniform = it == 2
        if uniform:
            lens = np.full(n, int(rng.choice([16, 32, 48, 256])))
        else:
            lens = [rng.integers(0, 60, size=n), np.where(rng.random(n) < 0.97, rng.integers(0, 90, size=n), rng.integers(500, 9000, size=n)),
                    rng.choice([0, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025], size=n)][int(rng.integers(0, 3))]
        so = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=so[1:])
        m = int(so[-1])
        ty = rng.integers(0, len(alg.desc) + (2 if rng.random() < 0.3 else 0), m)
        ev = S.make_events(ty, rng.integers(0, 1 << 30, m), rng.integers(-(1 << 31), 1 << 31, m))
        prior = None
        if rng.random() < 0.6:
            pl = rng.integers(0, 4, size=n)
            po = np.zeros(n + 1, np.int64)
            np.cumsum(pl, out=po[1:])
            pe = S.make_events(rng.integers(0, len(alg.desc), int(po[-1])), rng.integers(0, 1 << 30, int(po[-1])), rng.integers(-1000, 1000, int(po[-1])))
            prior = oracle.fold_csr(po, pe, None, alg)
            if rng.random() < 0.6:
                live = (prior["flags"] & S.STATE_PRESENT) != 0
                for f in ("count", "version", "min_arg", "max_arg"):
                    prior[f][live] = rng.integers(-1 << 31, 1 << 31, int(live.sum()))
                prior["sum64"][live] = rng.integers(-1 << 62, 1 << 62, int(live.sum()))
                prior["event_count"][live] = rng.integers(0, 1 << 32, int(live.sum()))
                prior["balance"][live] = rng.standard_normal(int(live.sum()))
        exp = oracle.fold_csr(so, ev, prior, alg)
        monkeypatch.setenv("SURGE_REPLAY_CHUNK_T", str(int(rng.choice([16, 64, 256, 4096]))))
        for build in ("1", "0"):
            monkeypatch.setenv("SURGE_REPLAY_RTC_LANES", build)
            for le in ("8", "16"):
                for name in ("SURGE_REPLAY_LE_SORTED", "SURGE_REPLAY_LE_CHUNKED", "SURGE_REPLAY_LE_ROWS"):
                    monkeypatch.setenv(name, le)
                with ReplayEngine(alg) as eng:
                    eng.load_csr(so, ev, prior)
                    for algo in [S.ALGO_SORTED, S.ALGO_CHUNKED] + ([S.ALGO_ROWS] if uniform and m else []):
                        eng.fold(algo)
                        assert eng.snapshot().tobytes() == exp.tobytes(), (seed, it, build, le, algo, alg.desc)
                    detail = eng.kernel_info()["detail"]
                    assert detail.startswith("lane kernels compiled for the op table") == (build == "1"), detail


@pytest.mark.parametrize("shape", ["zipf", "short", "long_tail", "uniform"])
def test_the_counting_sort_of_the_index_orders_rows_exactly_like_the_radix_sort(shape, monkeypatch):
    """the per-log index (length order of SORTED, row order of the chunk table) is built by a hand-written
    stable counting sort (length_sort.hip: per-wave LDS histograms, ballot ranking, no atomics) whenever the longest row is
    below 8192 events; rocPRIM's radix sort stays for longer rows.  Both are stable and descending: the two orders must be
    IDENTICAL element by element (SURGE_REPLAY_INDEX_SORT=radix forces the library sort), and so must the folds."""
    rng = np.random.default_rng(77)
    n = 40_000
    if shape == "zipf":
        lens = synth.zipf_lengths(np.arange(n, dtype=np.int64), 3)
    elif shape == "short":
        lens = rng.integers(0, 5, size=n)
    elif shape == "long_tail":
        lens = np.where(rng.random(n) < 0.999, rng.integers(1, 200, size=n), rng.integers(5000, 8100, size=n))
    else:
        lens = np.full(n, 48)
    so, ev = synth.csr_log(lens.astype(np.int64), 11, synth.STRESS_MIX)
    exp = oracle.fold_csr(so, ev)
    orders = {}
    for sort in ("counting", "radix"):
        if sort == "radix":
            monkeypatch.setenv("SURGE_REPLAY_INDEX_SORT", "radix")
        for chunk_t in ("4096", "256"):
            monkeypatch.setenv("SURGE_REPLAY_CHUNK_T", chunk_t)
            with ReplayEngine() as eng:
                eng.load_csr(so, ev)
                for algo in (S.ALGO_SORTED, S.ALGO_CHUNKED, S.ALGO_TILED):
                    eng.fold(algo)
                    assert eng.snapshot().tobytes() == exp.tobytes(), (shape, sort, chunk_t, algo)
                orders[(sort, chunk_t)] = (eng.index_order(S.ALGO_SORTED), eng.index_order(S.ALGO_CHUNKED))
    for chunk_t in ("4096", "256"):
        a, b = orders[("counting", chunk_t)], orders[("radix", chunk_t)]
        assert np.array_equal(a[0], b[0]), (shape, chunk_t, "length order")
        assert np.array_equal(a[1], b[1]), (shape, chunk_t, "chunk rows")
    # the length order itself: descending counts, equal counts in aggregate order (over the non-empty aggregates)
    nz = np.flatnonzero(lens > 0)
    ref = nz[np.argsort(-lens[nz], kind="stable")]
    got = orders[("counting", "4096")][0]
    assert np.array_equal(nz[got] if lens.min() == 0 else got, ref), shape


@pytest.mark.gpu
def test_short_rows_kernel_on_logs_of_many_short_aggregates_and_auto_picks_it():
    """SURGE_ALGO_SHORT (round 6): one lane per aggregate straight from the CSR arrays — the shape of a packed events topic whose
    aggregates published a handful of events each (the e2e topic: 10 M aggregates, 1.4 events each).  Rows of 0 .. 64 events
    with empties in runs and at both ends, a prior snapshot, throwing / deleting events; AUTO picks it for such a log (and not
    for a uniform one or one with a long aggregate); states = the oracle's, byte for byte."""
    rng = np.random.default_rng(9)
    n = 150_000
    lens = np.where(rng.random(n) < 0.3, 0, rng.integers(1, 9, size=n))
    lens[:300] = 0
    lens[-77:] = 0
    lens[1000] = 64
    so, ev = synth.csr_log(lens.astype(np.int64), 13, synth.STRESS_MIX)
    prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 3, size=n), 14, synth.STRESS_MIX))
    for init in (None, prior):
        exp = oracle.fold_csr(so, ev, init)
        got, st = gpu_fold(so, ev, init, algo=S.ALGO_SHORT)
        assert got.tobytes() == exp.tobytes() and st.last_algo == S.ALGO_SHORT
        got, st = gpu_fold(so, ev, init, algo=S.ALGO_AUTO)
        assert got.tobytes() == exp.tobytes() and st.last_algo == S.ALGO_SHORT
    lens[5] = 65  # one aggregate longer than the kernel is meant for: AUTO goes back to the flat fold
    so, ev = synth.csr_log(lens.astype(np.int64), 13, synth.STRESS_MIX)
    got, st = gpu_fold(so, ev, None, algo=S.ALGO_AUTO)
    assert st.last_algo == S.ALGO_FLAT and got.tobytes() == oracle.fold_csr(so, ev).tobytes()
    got, st = gpu_fold(so, ev, None, algo=S.ALGO_SHORT)  # asked for, it still folds any CSR
    assert got.tobytes() == oracle.fold_csr(so, ev).tobytes()
