"""The host mirror of the plugin surface: literal ``handleEvent`` code vs the declared event algebra.

``handle_event`` in surge_amd/fixtures.py is a case-by-case restatement of the Scala fixtures; the
``event_algebra`` beside it is what the kernels replay.  Folding random event sequences through
both (Python ``foldLeft`` vs the descriptor-driven oracle) ties the declaration to the code the
reference's own specs exercise.
"""
import random
import uuid

import numpy as np
import pytest

from oracle import oracle
from surge_amd import schema as S
from surge_amd.core import SurgeContext
from surge_amd.fixtures import (
    BankAccount, BankAccountCommandModel, BankAccountCreated, BankAccountUpdated, CounterBusinessLogic,
    CounterCommandModel, CountDecremented, CountIncremented, CreateAccount, CreateNoOpEvent, CreditAccount,
    DebitAccount, DoNothing, ExceptionThrowingEvent, Increment, NoOpEvent, State,
    AccountDoesNotExistException, InsufficientFundsException,
)


def fold_left(model, state, events):
    for e in events:  # events.foldLeft(state)(handleEvent) — CommandModels.scala:26
        state = model.handle_event(state, e)
    return state


def random_counter_events(rng, agg_id, n):
    out = []
    for i in range(n):
        r = rng.random()
        k = rng.choice([0, 1, 7, 2**31 - 1, -(2**31), rng.randrange(-2**31, 2**31)])
        if r < 0.45:
            out.append(CountIncremented(agg_id, k, i + 1))
        elif r < 0.9:
            out.append(CountDecremented(agg_id, k, i + 1))
        else:
            out.append(NoOpEvent(agg_id, i + 1))
    return out


@pytest.mark.parametrize("seed", range(5))
def test_counter_handle_event_equals_declared_algebra(seed):
    rng = random.Random(seed)
    model = CounterCommandModel()
    for start in (None, State("agg", 3, 3)):
        events = random_counter_events(rng, "agg", rng.randrange(0, 60))
        expect = fold_left(model, start, events)
        init = None if start is None else model.state_to_fixed(start)
        enc = model.encode_events(events)
        got = oracle.fold_csr(np.array([0, len(events)], dtype=np.int64), enc, init, model.event_algebra())[0]
        if expect is None:
            assert not got["flags"] & S.STATE_PRESENT
        else:
            assert got["flags"] == S.STATE_PRESENT
            assert model.state_from_fixed("agg", got) == expect


@pytest.mark.parametrize("seed", range(5))
def test_bank_account_handle_event_equals_declared_algebra(seed):
    rng = random.Random(100 + seed)
    model = BankAccountCommandModel()
    acct = uuid.UUID(int=seed + 1)
    events = []
    for _ in range(rng.randrange(0, 40)):
        if rng.random() < 0.2:
            events.append(BankAccountCreated(acct, "Jane Doe", "1234", rng.uniform(-1e6, 1e6)))
        else:
            events.append(BankAccountUpdated(acct, rng.uniform(-1e6, 1e6)))
    expect = fold_left(model, None, events)
    enc = model.encode_events(events)
    got = oracle.fold_csr(np.array([0, len(events)], dtype=np.int64), enc, None, model.event_algebra())[0]
    if expect is None:
        assert got.tobytes() == S.empty_states(1)[0].tobytes()
    else:
        assert model.state_from_fixed(str(acct), got) == expect  # f64 is SET-only => exact equality


# ---- the reference's own spec flows, through toCore (PersistentActorSpec.scala) --------------------
def test_increment_command_persists_event_and_updates_state():
    # PersistentActorSpec.scala:134-168
    core = CounterCommandModel().to_core()
    base = State("agg-1", 3, 3)
    ctx = core.handle(SurgeContext(state=base), base, Increment("agg-1"))
    assert [e for e, _ in ctx.events] == [CountIncremented("agg-1", 1, 4)]
    assert ctx.state == State("agg-1", 4, 4)
    assert ctx.replies[0](ctx.state) == State("agg-1", 4, 4)


def test_do_nothing_persists_nothing():
    # PersistentActorSpec.scala:229-273
    core = CounterCommandModel().to_core()
    base = State("agg-1", 3, 3)
    ctx = core.handle(SurgeContext(state=base), base, DoNothing("agg-1"))
    assert ctx.events == () and ctx.state == base


def test_noop_event_is_published_but_state_is_unchanged():
    # PersistentActorSpec.scala:495-508
    core = CounterCommandModel().to_core()
    base = State("agg-1", 3, 3)
    ctx = core.handle(SurgeContext(state=base), base, CreateNoOpEvent("agg-1"))
    assert [e for e, _ in ctx.events] == [NoOpEvent("agg-1", 4)] and ctx.state == base


def test_apply_events_folds_left_over_the_accumulator():
    # CommandModels.scala:26 (NOT the fixture quirk at core TestBoundedContext.scala:96)
    core = CounterCommandModel().to_core()
    base = State("agg-1", 3, 3)
    ctx = core.apply_async(SurgeContext(state=base), base, [CountIncremented("agg-1", 1, 4), CountIncremented("agg-1", 1, 5)])
    assert ctx.state == State("agg-1", 5, 5)


def test_throwing_event_propagates_like_a_failed_future():
    # PersistentActorSpec.scala:431-464
    core = CounterCommandModel().to_core()
    base = State("agg-1", 3, 3)
    with pytest.raises(RuntimeError, match="failed"):
        core.apply_async(SurgeContext(state=base), base, [ExceptionThrowingEvent("agg-1", 4, RuntimeError("failed"))])


def test_bank_account_commands():
    # BankAccountCommandEngineSpec.scala:44-68 + BankAccountCommandModel.scala:53-79
    m = BankAccountCommandModel()
    core = m.to_core()
    acct = uuid.uuid4()
    ctx = core.handle(SurgeContext(), None, CreateAccount(acct, "Jane Doe", "1234", 1000.0))
    assert ctx.state == BankAccount(acct, "Jane Doe", "1234", 1000.0)
    ctx2 = core.handle(SurgeContext(state=ctx.state), ctx.state, CreditAccount(acct, 100.0))
    assert ctx2.state.balance == 1100.0
    assert core.handle(SurgeContext(), ctx.state, CreateAccount(acct, "x", "y", 1.0)).events == ()
    with pytest.raises(AccountDoesNotExistException):
        core.handle(SurgeContext(), None, CreditAccount(acct, 1.0))
    with pytest.raises(InsufficientFundsException):
        core.handle(SurgeContext(), ctx.state, DebitAccount(acct, 1e9))


def test_formats_round_trip_and_match_oracle_json_text():
    bl = CounterBusinessLogic()
    st = State("stateKey1", 4, 4)
    ser = bl.aggregate_write_formatting().write_state(st)
    assert ser.value == oracle.counter_state_json("stateKey1", 4, 4)
    assert bl.aggregate_read_formatting().read_state(ser.value) == st
    assert bl.aggregate_read_formatting().read_state(b"not json") is None  # asOpt
    msg = bl.event_write_formatting().write_event(CountIncremented("stateKey1", 1, 4))
    assert msg.key == "stateKey1:4"  # TestBoundedContext.scala:123
    assert bl.event_write_formatting().read_event(msg) == CountIncremented("stateKey1", 1, 4)
