"""ABI v2 slot schemas (include/surge_replay.h): models the seven named fields of v1 cannot express.

* ``TwoCounters``: two Int counters and a Long version — three slots, four event types.
* ``Ledger``: an f64 balance that ACCUMULATES (credit / debit as ADD / SUB of the payload), a running max, a
  transaction count — the fold order matters for the doubles and must be the JVM's: strictly left to right.
The literal Python ``handle_event`` of each model (what the Scala would be) is the semantic contract; the C oracle's
slot interpreter and the GPU kernel (fold_slots.hip) are held to it and to each other, byte for byte."""
from dataclasses import dataclass, replace
from typing import Optional

import numpy as np
import pytest

from oracle import oracle
from surge_amd import schema as S
from surge_amd.schema import (CLS_CREATE, CLS_DELETE, CLS_MATERIALIZE, CLS_REQUIRE, D_POISON, OP_ADD, OP_MAX, OP_SET, OP_SUB, SLOT_F64,
                              SLOT_I32, SLOT_I64, SRC_ARG, SRC_ONE, SRC_PAYLOAD, SRC_SEQ, Slot, SlotAlgebra)

I32 = lambda x: ((int(x) + 2**31) % 2**32) - 2**31  # noqa: E731  JVM Int wrap


# ---- model 1: two counters + Long version ---------------------------------------------------------------------------
@dataclass(frozen=True)
class TwoCounters:
    a: int
    b: int
    version: int  # Long


TC_INC_A, TC_INC_B, TC_RESET_B, TC_THROW = 0, 1, 2, 3
TWO_COUNTERS = SlotAlgebra(
    slots=(Slot("a", SLOT_I32, SRC_ARG), Slot("b", SLOT_I32, SRC_ARG), Slot("version", SLOT_I64, SRC_SEQ)),
    types=(
        (CLS_MATERIALIZE, {"a": OP_ADD, "version": OP_SET}),          # AIncremented(by, seq)
        (CLS_MATERIALIZE, {"b": OP_SUB, "version": OP_SET}),          # BDecremented(by, seq)
        (CLS_REQUIRE, {"b": OP_SET}),                                 # BReset(to): aggregate.map(_.copy(b = to))
        (D_POISON, {}),
    ),
)


def two_counters_handle_event(agg: Optional[TwoCounters], ty, seq, arg) -> Optional[TwoCounters]:
    if ty == TC_THROW:
        raise RuntimeError("boom")
    if ty == TC_RESET_B:
        return None if agg is None else replace(agg, b=I32(arg))
    cur = agg if agg is not None else TwoCounters(0, 0, 0)
    if ty == TC_INC_A:
        return replace(cur, a=I32(cur.a + arg), version=seq)
    return replace(cur, b=I32(cur.b - arg), version=seq)


# ---- model 2: an accumulating f64 ledger ----------------------------------------------------------------------------
@dataclass(frozen=True)
class Ledger:
    balance: float
    largest: float
    transactions: int


LG_OPEN, LG_CREDIT, LG_DEBIT, LG_CLOSE = 0, 1, 2, 3
LEDGER = SlotAlgebra(
    slots=(Slot("balance", SLOT_F64, SRC_PAYLOAD), Slot("largest", SLOT_F64, SRC_PAYLOAD, default=float("-inf")),
           Slot("transactions", SLOT_I32, SRC_ONE)),
    types=(
        (CLS_CREATE, {"balance": OP_SET}),                                            # Opened(initial)
        (CLS_REQUIRE, {"balance": OP_ADD, "largest": OP_MAX, "transactions": OP_ADD}),  # Credited(amount)
        (CLS_REQUIRE, {"balance": OP_SUB, "largest": OP_MAX, "transactions": OP_ADD}),  # Debited(amount)
        (CLS_DELETE, {}),                                                             # Closed
    ),
    count_events=True,
)


def ledger_handle_event(agg: Optional[Ledger], ty, amount) -> Optional[Ledger]:
    if ty == LG_OPEN:
        return Ledger(amount, float("-inf"), 0)
    if ty == LG_CLOSE:
        return None
    if agg is None:
        return None
    bal = agg.balance + amount if ty == LG_CREDIT else agg.balance - amount  # IEEE double, in event order
    return Ledger(bal, amount if amount > agg.largest else agg.largest, I32(agg.transactions + 1))


def make_log(rng, n_agg, max_len, types, p, f64_types=()):
    lens = rng.integers(0, max_len, size=n_agg) * (rng.random(n_agg) < 0.9)
    so = np.zeros(n_agg + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    n = int(so[-1])
    ev = np.zeros(n, dtype=S.EVENT_DTYPE)
    ev["type"] = rng.choice(types, size=n, p=p)
    ev["seq"] = rng.integers(-(1 << 31), 1 << 31, size=n)
    ints = rng.integers(-(1 << 31), 1 << 31, size=n).astype(np.int64) & 0xFFFFFFFF
    # amounts that do not sum exactly: the order of the additions shows in the last bits
    dbl = (rng.random(n) * 10.0 ** rng.integers(-3, 9, size=n)).astype(np.float64).view(np.int64)
    ev["raw"] = np.where(np.isin(ev["type"], f64_types), dbl, ints).astype(np.uint64)
    return so, ev


def test_oracle_slot_interpreter_equals_the_literal_two_counters_model():
    rng = np.random.default_rng(1)
    so, ev = make_log(rng, 300, 40, [TC_INC_A, TC_INC_B, TC_RESET_B, TC_THROW], [0.45, 0.4, 0.14, 0.01])
    got = oracle.fold_csr_v2(so, ev, TWO_COUNTERS)
    for a in range(300):
        agg, poisoned = None, False
        for e in ev[so[a]:so[a + 1]]:
            try:
                agg = two_counters_handle_event(agg, int(e["type"]), int(e["seq"]), int(np.int32(np.uint32(e["raw"] & 0xFFFFFFFF))))
            except RuntimeError:
                poisoned = True
                break
        st = got[a]
        assert bool(st["flags"] & S.STATE_POISONED) == poisoned
        if agg is None:
            assert not st["flags"] & S.STATE_PRESENT and st["a"] == 0 and st["b"] == 0 and st["version"] == 0
        else:
            assert st["flags"] & S.STATE_PRESENT and (int(st["a"]), int(st["b"]), int(st["version"])) == (agg.a, agg.b, agg.version)


def test_oracle_slot_interpreter_equals_the_literal_ledger_and_keeps_the_addition_order():
    rng = np.random.default_rng(2)
    so, ev = make_log(rng, 200, 60, [LG_OPEN, LG_CREDIT, LG_DEBIT, LG_CLOSE], [0.08, 0.5, 0.4, 0.02], f64_types=[LG_OPEN, LG_CREDIT, LG_DEBIT])
    got = oracle.fold_csr_v2(so, ev, LEDGER)
    order_matters = 0
    for a in range(200):
        agg, n_applied = None, 0
        seg = ev[so[a]:so[a + 1]]
        for e in seg:
            was = agg
            agg = ledger_handle_event(agg, int(e["type"]), float(np.uint64(e["raw"]).view(np.float64)))
            if int(e["type"]) == LG_OPEN:
                n_applied = 1
            elif agg is not None and was is not None:
                n_applied += 1
        st = got[a]
        if agg is None:
            assert not st["flags"] & S.STATE_PRESENT
        else:
            assert st["balance"].tobytes() == np.float64(agg.balance).tobytes()  # bit for bit: same order of additions
            assert st["largest"] == agg.largest and int(st["transactions"]) == agg.transactions and int(st["event_count"]) == n_applied
            amounts = [float(np.uint64(e["raw"]).view(np.float64)) for e in seg]
            if len(amounts) > 3 and np.float64(agg.balance) != np.float64(sum(sorted(amounts))):
                order_matters += 1
    assert order_matters > 20  # the data really is order-sensitive (a reassociated sum would not pass the check above)


def test_slot_state_layout_and_schema_validation():
    import ctypes

    from surge_amd import _native

    dt = LEDGER.state_dtype()
    assert dt.itemsize == 64 and dt.fields["balance"][1] == 0 and dt.fields["transactions"][1] == 16 and dt.fields["flags"][1] == 36
    assert [S.slot_offset(i) for i in range(7)] == [0, 8, 16, 24, 40, 48, 56]
    lib = _native.load()
    h = ctypes.c_void_p()
    bad = TWO_COUNTERS.to_c()
    bad.ops[0] |= 9 << 4  # no such operation
    assert lib.surge_replay_create_v2(ctypes.byref(bad), 0, ctypes.byref(h)) == -5
    bad = TWO_COUNTERS.to_c()
    bad.ops[0] |= 1 << 24  # operation on slot 6, which the schema does not declare
    assert lib.surge_replay_create_v2(ctypes.byref(bad), 0, ctypes.byref(h)) == -1
    bad = TWO_COUNTERS.to_c()
    bad.abi_version = 1
    assert lib.surge_replay_create_v2(ctypes.byref(bad), 0, ctypes.byref(h)) == -5


# ---- java.lang.Math.min / max for F64 slots -------------------------------------------------------------------------
def java_math_min(a: float, b: float) -> float:
    """``java.lang.Math.min(double, double)`` as the JDK library source states it."""
    if a != a:
        return a
    if a == 0.0 and b == 0.0 and np.float64(b).view(np.uint64) == 0x8000000000000000:
        return b
    return a if a <= b else b


def java_math_max(a: float, b: float) -> float:
    if a != a:
        return a
    if a == 0.0 and b == 0.0 and np.float64(a).view(np.uint64) == 0x8000000000000000:
        return b
    return a if a >= b else b


EXTREMES = SlotAlgebra(
    slots=(Slot("lowest", SLOT_F64, SRC_PAYLOAD, default=float("inf")), Slot("highest", SLOT_F64, SRC_PAYLOAD, default=float("-inf"))),
    types=((CLS_MATERIALIZE, {"lowest": S.OP_MIN, "highest": OP_MAX}),),
)
SPECIAL_DOUBLES = np.array([0x0000000000000000, 0x8000000000000000, 0x7FF8000000000000, 0xFFF8000000000001, 0x7FF0000000000000,
                            0xFFF0000000000000, 0x3FF0000000000000, 0xBFF0000000000000, 0x0000000000000001, 0x7FF4000000000000],
                           dtype=np.uint64)  # +0 -0 NaN NaN(payload, sign) +inf -inf 1 -1 denormal sNaN


def extremes_log(rng, n_agg, max_len):
    lens = rng.integers(1, max_len, size=n_agg)
    so = np.zeros(n_agg + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    ev = np.zeros(int(so[-1]), dtype=S.EVENT_DTYPE)
    ev["raw"] = np.where(rng.random(ev.shape[0]) < 0.7, rng.choice(SPECIAL_DOUBLES, size=ev.shape[0]),
                         (rng.standard_normal(ev.shape[0]) * 3).view(np.uint64))
    return so, ev


def test_oracle_f64_min_max_are_java_math_min_max_including_nan_and_signed_zero():
    rng = np.random.default_rng(11)
    so, ev = extremes_log(rng, 400, 12)
    got = oracle.fold_csr_v2(so, ev, EXTREMES)
    saw_nan = saw_negzero = 0
    for a in range(400):
        lo, hi = float("inf"), float("-inf")
        for raw in ev["raw"][so[a]:so[a + 1]]:
            x = float(np.uint64(raw).view(np.float64))
            lo, hi = java_math_min(lo, x), java_math_max(hi, x)
        assert got[a]["lowest"].tobytes() == np.float64(lo).tobytes() and got[a]["highest"].tobytes() == np.float64(hi).tobytes()
        saw_nan += lo != lo
        saw_negzero += np.float64(lo).view(np.uint64) == 0x8000000000000000
    assert saw_nan > 20 and saw_negzero > 0  # the log really exercises both rules


def test_slot_schema_compiles_to_a_gfx950_code_object_without_a_gpu():
    """surge_replay_compile_schema_v2: the schema-specialised kernels are the interpreter's own device source compiled
    by hiprtc with the schema as constants — checkable on a build machine."""
    import ctypes

    from surge_amd import _native

    lib = _native.load()
    for algebra in (LEDGER, TWO_COUNTERS, EXTREMES):
        sc = algebra.to_c()
        n = ctypes.c_int64()
        rc = lib.surge_replay_compile_schema_v2(ctypes.byref(sc), b"gfx950", None, 0, ctypes.byref(n))
        assert rc == 0, lib.surge_replay_last_error(None).decode()
        buf = ctypes.create_string_buffer(n.value)
        assert lib.surge_replay_compile_schema_v2(ctypes.byref(sc), b"gfx950", buf, n.value, ctypes.byref(n)) == 0
        assert buf.raw[:4] == b"\x7fELF" and all(name in buf.raw for name in (b"surge_slots_csr8", b"surge_slots_csr16", b"surge_slots_tiled1", b"surge_slots_tiled2"))
        assert lib.surge_replay_compile_schema_v2(ctypes.byref(sc), b"gfx950", buf, 16, ctypes.byref(n)) == -1  # too small
    bad = LEDGER.to_c()
    bad.ops[0] |= 9 << 4
    assert lib.surge_replay_compile_schema_v2(ctypes.byref(bad), b"gfx950", None, 0, ctypes.byref(n)) == -5


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=["specialised", "interpreter"])
def kernel_build(request, monkeypatch):
    """Every GPU test of the slot fold runs against both builds of the same device code."""
    monkeypatch.setenv("SURGE_REPLAY_RTC", "1" if request.param == "specialised" else "0")
    return request.param


def _check_build(eng, kernel_build):
    info = eng.kernel_info()
    assert info["specialised"] == (kernel_build == "specialised"), info


@pytest.mark.gpu
@pytest.mark.parametrize("algebra,types,p,f64", [
    (TWO_COUNTERS, [TC_INC_A, TC_INC_B, TC_RESET_B, TC_THROW], [0.45, 0.4, 0.149, 0.001], ()),
    (LEDGER, [LG_OPEN, LG_CREDIT, LG_DEBIT, LG_CLOSE], [0.03, 0.55, 0.415, 0.005], (LG_OPEN, LG_CREDIT, LG_DEBIT)),
])
def test_gpu_slot_fold_is_bit_identical_to_the_sequential_oracle(algebra, types, p, f64, kernel_build):
    from surge_amd.replay import ReplayEngine, ReplayError

    rng = np.random.default_rng(3)
    with ReplayEngine(algebra) as eng:
        _check_build(eng, kernel_build)
        for n_agg, max_len in ((1, 5), (70, 3), (500, 700), (3000, 90), (40, 9000)):
            so, ev = make_log(rng, n_agg, max_len, types, p, f64)
            exp = oracle.fold_csr_v2(so, ev, algebra)
            eng.load_csr(so, ev)
            eng.fold()
            assert eng.stats().last_algo == S.ALGO_SLOTS
            got = eng.snapshot()
            assert got.tobytes() == exp.tobytes(), (n_agg, max_len)
            eng.fold(S.ALGO_TILED)  # the same log through the tile-major copy (whole aggregates as rows)
            assert eng.stats().last_algo == S.ALGO_TILED and eng.layout_info().cut_aggregates == 0
            assert eng.snapshot().tobytes() == exp.tobytes(), ("tiled", n_agg, max_len)
            # onto a prior snapshot, and K3 micro-batches onto the resident state (same kernel, one lane per group)
            so2, ev2 = make_log(rng, n_agg, max(2, max_len // 3), types, p, f64)
            eng.load_csr(so2, ev2, exp)
            eng.fold(S.ALGO_TILED)
            exp2 = oracle.fold_csr_v2(so2, ev2, algebra, exp)
            assert eng.snapshot().tobytes() == exp2.tobytes()
            eng.fold()
            assert eng.snapshot().tobytes() == exp2.tobytes()
            m = 4 * n_agg + 3
            agg_idx = rng.integers(0, n_agg, size=m)
            _, be = make_log(rng, 1, 2, types, p, f64)
            be = np.resize(make_log(rng, m, 3, types, p, f64)[1], m) if m else be
            eng.append_events(agg_idx, be)
            order = np.argsort(agg_idx, kind="stable")
            off = np.zeros(n_agg + 1, np.int64)
            np.cumsum(np.bincount(agg_idx, minlength=n_agg), out=off[1:])
            exp3 = oracle.fold_csr_v2(off, be[order], algebra, exp2)
            assert eng.snapshot().tobytes() == exp3.tobytes()
        with pytest.raises(ReplayError):
            eng.fold(S.ALGO_FLAT)  # a slot schema promises no associativity: only the one-lane-per-aggregate kernel
    with ReplayEngine() as v1, pytest.raises(ReplayError):
        v1.load_csr(*make_log(rng, 5, 5, [0, 1], [0.5, 0.5]))
        v1.fold(S.ALGO_SLOTS)


@pytest.mark.gpu
def test_gpu_f64_min_max_follow_java_math_on_nan_and_signed_zero(kernel_build):
    from surge_amd.replay import ReplayEngine

    rng = np.random.default_rng(12)
    so, ev = extremes_log(rng, 5000, 40)
    exp = oracle.fold_csr_v2(so, ev, EXTREMES)
    with ReplayEngine(EXTREMES) as eng:
        _check_build(eng, kernel_build)
        eng.load_csr(so, ev)
        for algo in (S.ALGO_AUTO, S.ALGO_TILED):
            eng.fold(algo)
            assert eng.snapshot().tobytes() == exp.tobytes(), algo


def random_slot_algebra(rng):
    """Any mix the ABI allows: 1-7 slots of any type / operand source, 2-16 event types of any class with any operation
    per slot, random defaults."""
    n_slots, n_types = int(rng.integers(1, 8)), int(rng.integers(2, 17))
    slots = []
    for i in range(n_slots):
        ty = int(rng.choice([SLOT_I32, SLOT_I64, SLOT_F64]))
        default = float(rng.choice([0.0, -0.0, 1.5, float("inf")])) if ty == SLOT_F64 else int(rng.integers(-5, 6))
        slots.append(Slot(f"s{i}", ty, int(rng.integers(0, 4)), default))
    types = []
    for _ in range(n_types):
        cls = int(rng.choice([CLS_MATERIALIZE, CLS_REQUIRE, CLS_CREATE, CLS_DELETE, D_POISON], p=[0.4, 0.3, 0.15, 0.1, 0.05]))
        types.append((cls, {f"s{i}": int(rng.integers(0, 6)) for i in range(n_slots) if rng.random() < 0.7}))
    return SlotAlgebra(slots=tuple(slots), types=tuple(types), count_events=bool(rng.integers(0, 2)))


@pytest.mark.gpu
def test_gpu_random_slot_schemas_both_builds_both_transports(monkeypatch):
    """Schema fuzz: the specialised kernels (one hiprtc compilation per schema), the interpreter, the CSR and the
    tile-major transport must all equal the sequential oracle, whatever the schema — incl. out-of-range event types
    (MatchError), NaN / signed-zero payloads, a prior snapshot, micro-batches."""
    from surge_amd.replay import ReplayEngine

    rng = np.random.default_rng(77)
    for case in range(6):
        algebra = random_slot_algebra(rng)
        n_types = len(algebra.types)
        n_agg = int(rng.integers(1, 1500))
        lens = rng.integers(0, int(rng.choice([4, 40, 400])), size=n_agg)
        so = np.zeros(n_agg + 1, np.int64)
        np.cumsum(lens, out=so[1:])
        n = int(so[-1])
        ev = np.zeros(n, dtype=S.EVENT_DTYPE)
        ev["type"] = np.where(rng.random(n) < 0.002, 99, rng.integers(0, n_types, size=n))
        ev["seq"] = rng.integers(-(1 << 31), 1 << 31, size=n)
        ev["raw"] = np.where(rng.random(n) < 0.2, rng.choice(SPECIAL_DOUBLES, size=n),
                             np.where(rng.random(n) < 0.5, (rng.standard_normal(n) * 1e3).view(np.uint64),
                                      rng.integers(0, 1 << 63, size=n).astype(np.uint64)))
        exp = oracle.fold_csr_v2(so, ev, algebra)
        exp2 = oracle.fold_csr_v2(so, ev, algebra, exp)
        for build in ("1", "0"):
            monkeypatch.setenv("SURGE_REPLAY_RTC", build)
            with ReplayEngine(algebra) as eng:
                assert eng.kernel_info()["specialised"] == (build == "1"), eng.kernel_info()
                eng.load_csr(so, ev)
                for algo in (S.ALGO_AUTO, S.ALGO_TILED):
                    eng.fold(algo)
                    assert eng.snapshot().tobytes() == exp.tobytes(), (case, build, algo)
                eng.load_csr(so, ev, exp)
                for algo in (S.ALGO_TILED, S.ALGO_AUTO):
                    eng.fold(algo)
                    assert eng.snapshot().tobytes() == exp2.tobytes(), (case, build, algo, "prior")
                if n:
                    m = min(n, 3000)
                    agg_idx = rng.integers(0, n_agg, size=m)
                    eng.append_events(agg_idx, ev[:m])
                    order = np.argsort(agg_idx, kind="stable")
                    off = np.zeros(n_agg + 1, np.int64)
                    np.cumsum(np.bincount(agg_idx, minlength=n_agg), out=off[1:])
                    assert eng.snapshot().tobytes() == oracle.fold_csr_v2(off, ev[:m][order], algebra, exp2).tobytes(), (case, build, "batch")


@pytest.mark.gpu
def test_gpu_slot_fold_mid_size_zipf_log(kernel_build):
    import torch

    from surge_amd import synth
    from surge_amd.replay import ReplayEngine

    n = 200_000
    lens = synth.zipf_lengths(np.arange(n, dtype=np.int64), 3, max_len=1024)
    so = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    rng = np.random.default_rng(9)
    ne = int(so[-1])
    ev = np.zeros(ne, dtype=S.EVENT_DTYPE)
    ev["type"] = rng.choice([LG_OPEN, LG_CREDIT, LG_DEBIT], size=ne, p=[0.02, 0.53, 0.45])
    ev["raw"] = (rng.random(ne) * 1e6).view(np.uint64)
    exp = oracle.fold_csr_v2(so, ev, LEDGER)
    with ReplayEngine(LEDGER) as eng:
        _check_build(eng, kernel_build)
        eng.load_csr(so, ev)
        for algo in (S.ALGO_AUTO, S.ALGO_TILED):
            eng.fold(algo)
            eng.fold(algo)
            got = eng.snapshot()
            st = eng.stats()
            assert got.tobytes() == exp.tobytes()
            print(f"slots kernel ({kernel_build}, algo {st.last_algo}): {ne} events, {st.last_fold_kernel_ms:.3f} ms, "
                  f"{st.algorithmic_bytes / st.last_fold_kernel_ms / 1e6:.0f} GB/s")
