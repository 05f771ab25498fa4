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


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("algebra,types,p,f64", [
    (TWO_COUNTERS, [TC_INC_A, TC_INC_B, TC_RESET_B, TC_THROW], [0.45, 0.4, 0.149, 0.001], ()),
    (LEDGER, [LG_OPEN, LG_CREDIT, LG_DEBIT, LG_CLOSE], [0.03, 0.55, 0.415, 0.005], (LG_OPEN, LG_CREDIT, LG_DEBIT)),
])
def test_gpu_slot_fold_is_bit_identical_to_the_sequential_oracle(algebra, types, p, f64):
    from surge_amd.replay import ReplayEngine, ReplayError

    rng = np.random.default_rng(3)
    with ReplayEngine(algebra) as eng:
        for n_agg, max_len in ((1, 5), (70, 3), (500, 700), (3000, 90), (40, 9000)):
            so, ev = make_log(rng, n_agg, max_len, types, p, f64)
            exp = oracle.fold_csr_v2(so, ev, algebra)
            eng.load_csr(so, ev)
            eng.fold()
            assert eng.stats().last_algo == S.ALGO_SLOTS
            got = eng.snapshot()
            assert got.tobytes() == exp.tobytes(), (n_agg, max_len)
            # onto a prior snapshot, and K3 micro-batches onto the resident state (same kernel, one lane per group)
            so2, ev2 = make_log(rng, n_agg, max(2, max_len // 3), types, p, f64)
            eng.load_csr(so2, ev2, exp)
            eng.fold()
            exp2 = oracle.fold_csr_v2(so2, ev2, algebra, exp)
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
def test_gpu_slot_fold_mid_size_zipf_log():
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
    with ReplayEngine(LEDGER) as eng:
        eng.load_csr(so, ev)
        eng.fold()
        got = eng.snapshot()
        st = eng.stats()
    exp = oracle.fold_csr_v2(so, ev, LEDGER)
    assert got.tobytes() == exp.tobytes()
    print(f"slots kernel: {ne} events, {st.last_fold_kernel_ms:.3f} ms, {st.algorithmic_bytes / st.last_fold_kernel_ms / 1e6:.0f} GB/s")
