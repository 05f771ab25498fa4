"""The host mirror of the plugin surface: literal ``handleEvent`` code vs the declared event algebra.

``handle_event`` in examples/fixture_models.py is a case-by-case restatement of the Scala fixtures; the
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
from fixture_models import (
    BankAccount, BankAccountCommandModel, BankAccountCreated, BankAccountUpdated, CounterBusinessLogic,
    CounterCommandModel, CountDecremented, CountIncremented, CreateAccount, CreateNoOpEvent, CreditAccount,
    DebitAccount, DoNothing, ExceptionThrowingEvent, Increment, NoOpEvent, State,
    AccountDoesNotExistException, InsufficientFundsException,
    DepositMoney, MoneyDeposited, SDK_SAMPLE_MODEL, SdkBankAccount, SdkEvent, SdkSampleCommandModel, sdk_sample_state_bytes,
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


@pytest.mark.parametrize("seed", range(4))
def test_sdk_sample_event_handler_equals_declared_algebra(seed):
    # R8: CQRSModel.applyEvents (scalasdk/Model.scala:11-13) over the sample's Int balance (sample/Main.scala:25-30)
    rng = random.Random(200 + seed)
    model = SdkSampleCommandModel()
    for start in (None, SdkBankAccount(41)):
        deposits = [MoneyDeposited(rng.choice([0, 1, 2**31 - 1, rng.randrange(0, 2**31)])) for _ in range(rng.randrange(0, 50))]
        expect = SDK_SAMPLE_MODEL.apply_events(start, deposits)
        events = [SdkEvent("acct", d) for d in deposits]
        assert fold_left(model, start, events) == expect
        init = None if start is None else model.state_to_fixed(start)
        got = oracle.fold_csr(np.array([0, len(events)], dtype=np.int64), model.encode_events(events), init, model.event_algebra())[0]
        if expect is None:
            assert not got["flags"] & S.STATE_PRESENT
        else:
            assert got["flags"] == S.STATE_PRESENT and model.state_from_fixed("acct", got) == expect


def test_sdk_sample_command_handler_and_stored_bytes():
    # sample/Main.scala:32-38: negative deposits are rejected (Left), the rest yield one event and the folded state
    assert SDK_SAMPLE_MODEL.execute_command(None, DepositMoney(-1)) == ("Left", "Amount cannot be < 0")
    assert SDK_SAMPLE_MODEL.execute_command(SdkBankAccount(3), DepositMoney(4)) == ("Right", ([MoneyDeposited(4)], SdkBankAccount(7)))
    assert SDK_SAMPLE_MODEL.execute_command(None, DepositMoney(0)) == ("Right", ([MoneyDeposited(0)], SdkBankAccount(0)))
    # stored form = protobuf State{aggregateId, payload = json4s text}: the fixture's bytes equal the protobuf runtime's
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    fd = descriptor_pb2.FileDescriptorProto(name="sdk_state.proto", syntax="proto3")
    m = fd.message_type.add(name="State")
    F = descriptor_pb2.FieldDescriptorProto
    m.field.add(name="aggregateId", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    m.field.add(name="payload", number=2, type=F.TYPE_BYTES, label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    State_pb = message_factory.GetMessageClass(pool.FindMessageTypeByName("State"))
    for agg_id, bal in [("0c3f1d9e-7a55-4a5c-9d5e-2f1f6f6f0001", 1100), ("", -5), ("k" * 200, 2**31 - 1)]:
        assert sdk_sample_state_bytes(agg_id, SdkBankAccount(bal)) == \
            State_pb(aggregateId=agg_id, payload=('{"balance":%d}' % bal).encode()).SerializeToString()


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


# ---- the other model flavours that compile to the same core (R4) -----------------------------------
def test_event_only_and_batched_models_fold_like_the_scala_command_model():
    import random

    from surge_amd.command import AggregateEventModel, BatchedAggregateCommandModel
    from surge_amd.core import SurgeContext
    from fixture_models import CounterCommandModel, CountDecremented, CountIncremented, Increment, NoOpEvent, State

    scala = CounterCommandModel()

    class EventOnly(AggregateEventModel):  # AggregateEventModel.scala:10-22
        def handle_events(self, state, events):
            for e in events:
                state = scala.handle_event(state, e)
            return state

    class Batched(BatchedAggregateCommandModel):  # AsyncAggregateCommandModel, CommandModels.scala:33-57
        def process_command(self, agg, cmd):
            return scala.process_command(agg, cmd)

        def handle_events(self, agg, events):
            return EventOnly().handle_events(agg, events)

    rng = random.Random(3)
    for _ in range(50):
        events = []
        for seq in range(1, rng.randrange(1, 30)):
            events.append(rng.choice([CountIncremented("a", rng.randrange(9), seq), CountDecremented("a", rng.randrange(9), seq), NoOpEvent("a", seq)]))
        start = rng.choice([None, State("a", 3, 3)])
        want = scala.to_core().apply_async(SurgeContext(state=start), start, events).state
        for model in (EventOnly(), Batched()):
            assert model.to_core().apply_async(SurgeContext(state=start), start, events).state == want
    # commands: the batched model persists the same events and reaches the same state; the event-only model refuses
    for model in (Batched(),):
        ctx = model.to_core().handle(SurgeContext(state=State("a", 3, 3)), State("a", 3, 3), Increment("a"))
        assert ctx.state == State("a", 4, 4) and [e for e, _ in ctx.events] == [CountIncremented("a", 1, 4)]
    with pytest.raises(NotImplementedError):
        EventOnly().to_core().handle(SurgeContext(), None, Increment("a"))
