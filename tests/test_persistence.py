"""The callers of the fold (R3 / R5 / R10 / R11 / N4) held to the literal expectations of the reference's own specs:
PersistentActorSpec.scala (:134-168, :229-308, :419-529), KafkaProducerActorImplSpec.scala (:301-368, :692-705),
MultilanguageGatewayServiceImplSpec.scala (:72-135) — multi-step SEQUENCES, the state of record on the GPU."""
import json

import pytest

from fixture_models import (
    CounterBusinessLogic, CountIncremented, CreateExceptionThrowingEvent, CreateNoOpEvent, Decrement, DoNothing,
    ExceptionThrowingEvent, FailCommandProcessing, Increment, State,
)
from surge_amd.persistence import ACKError, ACKSuccess, EventRecord, GpuPersistentActor, InFlightTracker, RetryConfig, StatePublisher
from surge_amd.snapshot import StateRecord
from surge_amd.store import AggregateInitializationException


# ---- CPU: the producer's in-flight bookkeeping (KafkaProducerActorImplSpec.scala:343-368, :692-705) ------------------
def test_is_aggregate_state_current_follows_ktable_progress():
    t = InFlightTracker()
    assert t.is_aggregate_state_current("bar")          # nothing published yet
    t.add_in_flight([("bar", 101)])                     # state record for key bar published at offset 101
    assert not t.is_aggregate_state_current("bar") and t.is_aggregate_state_current("foo")
    t.processed_up_to(100)                              # KTable is still behind
    assert not t.is_aggregate_state_current("bar")
    t.processed_up_to(101)                              # KTableProgressUpdate(LagInfo(101, 101))
    assert t.is_aggregate_state_current("bar")


def test_in_flight_keeps_only_the_newest_offset_per_key():
    t = InFlightTracker()
    t.add_in_flight([("a", 5), ("b", 6), ("a", 9), ("a", 7)])
    assert t.in_flight_for_aggregate("a") == [9] and t.in_flight_for_aggregate("b") == [6]
    t.processed_up_to(8)
    assert t.in_flight_for_aggregate("a") == [9] and t.in_flight_for_aggregate("b") == []


class _FakeStore:
    """S2 only (no GPU): what KTableInitializationSupport talks to."""

    def __init__(self, values, failures=0):
        self.values, self.failures, self.reads = values, failures, 0

    def get_aggregate_bytes(self, aggregate_id):
        self.reads += 1
        if self.failures > 0:
            self.failures -= 1
            raise IOError("InvalidStateStoreException: rebalancing")
        return self.values.get(aggregate_id)


def test_initialization_retries_then_succeeds_or_fails_like_ktable_initialization_support():
    bl = CounterBusinessLogic()
    fmt = bl.aggregate_write_formatting()
    slept = []
    # (a) two failed reads (2 s each), then the state arrives — KTableInitializationSupport.scala:63-81
    store = _FakeStore({"x": fmt.write_state(State("x", 3, 3)).value}, failures=2)
    actor = GpuPersistentActor(bl, "x", store, StatePublisher(store), sleep=slept.append)
    assert actor.get_state() == State("x", 3, 3) and slept == [2.0, 2.0] and actor.initialization_attempts == 2
    # (b) state not current in the KTable: 500 ms retries until the producer reports current — :37-61
    slept.clear()
    pub = StatePublisher(store)
    pub.tracker.add_in_flight([("x", 7)])
    calls = {"n": 0}

    def sleep(s):
        slept.append(s)
        calls["n"] += 1
        if calls["n"] == 3:
            pub.tracker.processed_up_to(7)

    actor = GpuPersistentActor(bl, "x", store, pub, sleep=sleep)
    assert actor.get_state() == State("x", 3, 3) and slept == [0.5, 0.5, 0.5]
    # (c) never current: max-initialization-attempts (10) exceeded -> AggregateInitializationException
    pub = StatePublisher(store)
    pub.tracker.add_in_flight([("x", 7)])
    actor = GpuPersistentActor(bl, "x", store, pub, retry=RetryConfig(max_initialization_attempts=10))
    with pytest.raises(AggregateInitializationException):
        actor.get_state()
    assert actor.initialization_attempts == 11
    # (d) a KTable miss is a valid initialisation: None
    actor = GpuPersistentActor(bl, "nobody", _FakeStore({}), StatePublisher(store))
    assert actor.get_state() is None


def test_a_rejected_command_is_an_ackrejection_not_an_ackerror_and_publishes_nothing():
    """PersistentActor.handle: `if (ctx.isRejected) ACKRejection(ctx.rejection)` (PersistentActor.scala:60-62, :205-207) —
    a reply type of its own, the state untouched, nothing handed to the producer."""
    from surge_amd.core import SurgeProcessingModel
    from surge_amd.persistence import ACKRejection

    bl = CounterBusinessLogic()
    fmt = bl.aggregate_write_formatting()
    store = _FakeStore({"x": fmt.write_state(State("x", 3, 3)).value})
    pub = StatePublisher(store)
    published = []
    pub.publish = lambda *a, **k: published.append(a)
    actor = GpuPersistentActor(bl, "x", store, pub)

    class Rejecting(SurgeProcessingModel):
        def handle(self, ctx, state, msg):
            return ctx.reject({"reason": "insufficient funds", "command": msg})

    actor.core = Rejecting()
    r = actor.process_message("withdraw 10")
    assert r == ACKRejection({"reason": "insufficient funds", "command": "withdraw 10"}) and not isinstance(r, ACKError)
    assert published == [] and actor.get_state() == State("x", 3, 3)


# ---- GPU: the spec's scenarios, state of record on the GPU store ----------------------------------------------------
def _context(base_events):
    from surge_amd.store import GpuReplayStateStore

    bl = CounterBusinessLogic()
    store = GpuReplayStateStore(bl)
    store.restore(base_events)
    return bl, store, StatePublisher(store)


def _base_3_3(agg="agg-1"):
    # TestContext.setupDefault: baseState = State(id, 3, 3) served by the (mock) KTable — here recovered from its events
    return [CountIncremented(agg, 1, 1), CountIncremented(agg, 1, 2), CountIncremented(agg, 1, 3)]


@pytest.mark.gpu
def test_properly_initialize_from_the_store_and_publish_event_plus_state():
    # PersistentActorSpec.scala:218-226 + processIncrementCommand :134-168
    bl, store, pub = _context(_base_3_3())
    try:
        actor = GpuPersistentActor(bl, "agg-1", store, pub, assigned_partition=1)
        assert actor.get_state() == State("agg-1", 3, 3)                       # GetState -> StateResponse(Some(baseState))
        assert actor.process_message(Increment("agg-1")) == ACKSuccess(State("agg-1", 4, 4))
        (batch,) = pub.published                                               # exactly one publish ...
        assert len(batch) == 2                                                 # ... of 2 records: the event and the state
        ev, st = batch
        assert isinstance(ev, EventRecord) and ev.topic == "testEventsTopic" and ev.key == "agg-1:4"
        assert json.loads(ev.value) == {"aggregateId": "agg-1", "incrementBy": 1, "sequenceNumber": 4, "_type": "countIncremented"}
        assert isinstance(st, StateRecord) and (st.topic, st.partition, st.key) == ("testStateTopic", 1, "agg-1")
        assert json.loads(st.value) == {"aggregateId": "agg-1", "count": 4, "version": 4}   # the spec compares parsed JSON (:157-161)
        # the KTable catches up: the event is folded onto the GPU state; the store now serves exactly the published bytes
        assert not pub.tracker.is_aggregate_state_current("agg-1")
        pub.ktable_progress()
        assert pub.tracker.is_aggregate_state_current("agg-1")
        assert store.get_aggregate_bytes("agg-1") == st.value
    finally:
        store.close()


@pytest.mark.gpu
def test_not_publish_when_nothing_changed():
    # :229-288 — DoNothing (no events); ApplyEvents[CountIncremented(id, 0, 3)] on (3,3): same state => nothing published
    bl, store, pub = _context(_base_3_3())
    try:
        actor = GpuPersistentActor(bl, "agg-1", store, pub)
        assert actor.process_message(DoNothing("agg-1")) == ACKSuccess(State("agg-1", 3, 3))
        assert pub.published == []
        assert actor.apply_events([CountIncremented("agg-1", 0, 3)]) == ACKSuccess(State("agg-1", 3, 3))
        assert pub.published == []
        # publishStateOnly = true changes nothing about "not publishing" (:246-261)
        bl.publish_state_only = True
        assert GpuPersistentActor(bl, "agg-1", store, pub).process_message(DoNothing("agg-1")) == ACKSuccess(State("agg-1", 3, 3))
        assert pub.published == []
    finally:
        store.close()


@pytest.mark.gpu
def test_publish_everything_or_only_the_state_change():
    # :292-330 — publishStateOnly false: 2 records; true: 1 record (the state)
    for state_only, n in ((False, 2), (True, 1)):
        bl, store, pub = _context(_base_3_3())
        try:
            bl.publish_state_only = state_only
            actor = GpuPersistentActor(bl, "agg-1", store, pub)
            assert actor.process_message(Increment("agg-1")) == ACKSuccess(State("agg-1", 4, 4))
            assert [len(b) for b in pub.published] == [n] and isinstance(pub.published[0][-1], StateRecord)
        finally:
            store.close()


@pytest.mark.gpu
def test_exceptions_from_the_domain_become_ackerror_and_leave_the_actor_usable():
    # :431-464
    bl, store, pub = _context(_base_3_3())
    try:
        actor = GpuPersistentActor(bl, "agg-1", store, pub)
        r = actor.process_message(FailCommandProcessing("agg-1", RuntimeError("failed")))
        assert isinstance(r, ACKError) and str(r.exception) == "failed"
        r = actor.process_message(CreateExceptionThrowingEvent("agg-1", RuntimeError("failed")))
        assert isinstance(r, ACKError) and str(r.exception) == "failed"
        r = actor.apply_events([ExceptionThrowingEvent("agg-1", 1, RuntimeError("failed"))])
        assert isinstance(r, ACKError) and str(r.exception) == "failed"
        assert pub.published == []
        assert actor.process_message(DoNothing("agg-1")) == ACKSuccess(State("agg-1", 3, 3))  # still (3,3), still usable
        pub.ktable_progress()
        assert store.get_aggregate("agg-1") == State("agg-1", 3, 3)
    finally:
        store.close()


@pytest.mark.gpu
def test_commands_one_at_a_time_noop_events_and_apply_events_sequences():
    bl, store, pub = _context(_base_3_3())
    try:
        actor = GpuPersistentActor(bl, "agg-1", store, pub)
        # "Process commands one at a time" :466-493 — two increments: (4,4) then (5,5)
        assert actor.process_message(Increment("agg-1")) == ACKSuccess(State("agg-1", 4, 4))
        assert actor.process_message(Increment("agg-1")) == ACKSuccess(State("agg-1", 5, 5))
        # "Publish events even if they don't update the state" :495-508 — the NoOp event is published, state unchanged
        n = len(pub.published)
        assert actor.process_message(CreateNoOpEvent("agg-1")) == ACKSuccess(State("agg-1", 5, 5))
        assert len(pub.published) == n + 1 and pub.published[-1][0].key == "agg-1:6"
        # "Handle ApplyEvent requests" :512-529 — two ApplyEvents of one event each; both publishes carry ONLY state records
        n = len(pub.published)
        assert actor.apply_events([CountIncremented("agg-1", 1, 6)]) == ACKSuccess(State("agg-1", 6, 6))
        assert actor.apply_events([CountIncremented("agg-1", 1, 7)]) == ACKSuccess(State("agg-1", 7, 7))
        assert [[type(r) for r in b] for b in pub.published[n:]] == [[StateRecord], [StateRecord]]
        # the KTable catches up with everything: GPU store == the last published state record, byte for byte
        pub.ktable_progress()
        assert store.get_aggregate_bytes("agg-1") == pub.published[-1][-1].value
        # a NEW actor for the aggregate (passivation / rebalance) initialises from the GPU store to the same state
        assert GpuPersistentActor(bl, "agg-1", store, pub).get_state() == State("agg-1", 7, 7)
    finally:
        store.close()


@pytest.mark.gpu
def test_multilanguage_gateway_sequence_on_a_new_aggregate():
    # MultilanguageGatewayServiceImplSpec.scala:72-135 — new aggregate + Increment => (1,1); Increment => (2,2); Decrement => (1,3)
    bl, store, pub = _context(_base_3_3("someone-else"))
    try:
        actor = GpuPersistentActor(bl, "fresh", store, pub)
        assert actor.get_state() is None
        assert actor.process_message(Increment("fresh")) == ACKSuccess(State("fresh", 1, 1))
        assert actor.process_message(Increment("fresh")) == ACKSuccess(State("fresh", 2, 2))
        assert actor.process_message(Decrement("fresh")) == ACKSuccess(State("fresh", 1, 3))
        pub.ktable_progress()  # the aggregate did not exist at recovery: the resident state grows (surge_replay_grow)
        assert store.get_aggregate("fresh") == State("fresh", 1, 3)
        assert store.get_aggregate("someone-else") == State("someone-else", 3, 3)
    finally:
        store.close()


# ---- CPU property test: the protocol over random command / ApplyEvents sequences, against a literal KTable ------------------
class _LiteralKTable:
    """S2 backed by the model's literal handle_event (no GPU): what the GPU store must be indistinguishable from."""

    def __init__(self, bl):
        self.bl, self.state = bl, {}

    def apply_events(self, events):
        model = self.bl.command_model()
        for e in events:
            self.state[e.aggregateId] = model.handle_event(self.state.get(e.aggregateId), e)

    def get_aggregate_bytes(self, aggregate_id):
        s = self.state.get(aggregate_id)
        return None if s is None else self.bl.aggregate_write_formatting().write_state(s).value


def test_random_sequences_keep_actor_store_and_published_state_records_in_step():
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings
    from hypothesis import strategies as st

    step = st.tuples(st.sampled_from(["a", "b", "c"]),
                     st.sampled_from(["inc", "dec", "nothing", "noop", "fail", "throwing", "apply", "apply0", "progress", "restart"]))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(step, min_size=1, max_size=40), st.booleans())
    def run(steps, state_only):
        bl = CounterBusinessLogic()
        bl.publish_state_only = state_only
        store = _LiteralKTable(bl)
        pub = StatePublisher(store)
        actors, expect = {}, {}

        def actor(agg):
            if agg not in actors:
                pub.ktable_progress()  # an actor may only initialise once the KTable has caught up
                actors[agg] = GpuPersistentActor(bl, agg, store, pub)
            return actors[agg]

        for agg, op in steps:
            before = len(pub.published)
            cur = expect.get(agg)
            version = cur.version if cur else 0
            if op == "progress":
                pub.ktable_progress()
                continue
            if op == "restart":  # passivation: the next message creates a fresh actor that re-reads the store
                actors.pop(agg, None)
                continue
            a = actor(agg)
            if op in ("inc", "dec"):
                r = a.process_message(Increment(agg) if op == "inc" else Decrement(agg))
                count = (cur.count if cur else 0) + (1 if op == "inc" else -1)
                expect[agg] = State(agg, count, version + 1)
                assert r == ACKSuccess(expect[agg]) and len(pub.published) == before + 1
                assert len(pub.published[-1]) == (1 if state_only else 2)
            elif op == "nothing":
                assert a.process_message(DoNothing(agg)) == ACKSuccess(cur) and len(pub.published) == before
            elif op == "noop":  # the event is published although the state may not change
                r = a.process_message(CreateNoOpEvent(agg))
                expect[agg] = cur if cur else State(agg, 0, 0)
                assert r == ACKSuccess(expect[agg]) and len(pub.published) == before + 1
            elif op == "fail":
                assert isinstance(a.process_message(FailCommandProcessing(agg, RuntimeError("x"))), ACKError) and len(pub.published) == before
            elif op == "throwing":
                assert isinstance(a.process_message(CreateExceptionThrowingEvent(agg, RuntimeError("x"))), ACKError) and len(pub.published) == before
            elif op == "apply":
                r = a.apply_events([CountIncremented(agg, 2, version + 1)])
                expect[agg] = State(agg, (cur.count if cur else 0) + 2, version + 1)
                assert r == ACKSuccess(expect[agg]) and [type(x) for x in pub.published[-1]] == [StateRecord]
            elif op == "apply0" and cur is not None:  # an event that leaves the state equal publishes nothing
                assert a.apply_events([CountIncremented(agg, 0, version)]) == ACKSuccess(cur) and len(pub.published) == before
            if len(pub.published) > before:  # the last record of every publish is the state the actor now holds
                last = pub.published[-1][-1]
                assert isinstance(last, StateRecord) and last.key == agg
                assert last.value == bl.aggregate_write_formatting().write_state(expect[agg]).value
        pub.ktable_progress()
        for agg, s in expect.items():  # the KTable (here literal; on the GPU: the resident state) ends where the actors are
            assert store.get_aggregate_bytes(agg) == (None if s is None else bl.aggregate_write_formatting().write_state(s).value)
            assert pub.tracker.is_aggregate_state_current(agg)

    run()


@pytest.mark.gpu
def test_bank_account_command_sequence_state_of_record_on_the_gpu():
    """The docs sample (BankAccountCommandModel.scala:53-86; the disabled BankAccountCommandEngineSpec.scala:44-68 holds the
    numbers: create 1000.0, credit 100.0 -> 1100.0) through the actor protocol: f64 balances are bit-copied by the fold
    (CREATE / REQUIRE classes), domain rejections publish nothing."""
    import uuid

    from surge_amd.command import SurgeCommandBusinessLogic
    from surge_amd.core import KafkaTopic
    from fixture_models import (
        AccountDoesNotExistException, BankAccount, BankAccountCommandModel, BankAccountFormat, CreateAccount, CreditAccount,
        DebitAccount, InsufficientFundsException,
    )
    from surge_amd.store import GpuReplayStateStore

    class BL(SurgeCommandBusinessLogic):
        aggregate_name = "BankAccount"
        state_topic, events_topic = KafkaTopic("bank-account-state"), KafkaTopic("bank-account-events")
        publish_state_only = True  # the sample publishes events too; their text (play-json Double) is parity-unpinned

        def __init__(self):
            self.m, self.f = BankAccountCommandModel(), BankAccountFormat()

        def command_model(self):
            return self.m

        def aggregate_read_formatting(self):
            return self.f

        def aggregate_write_formatting(self):
            return self.f

    bl = BL()
    store = GpuReplayStateStore(bl)
    try:
        other = uuid.UUID(int=99)
        from fixture_models import BankAccountCreated

        store.restore([BankAccountCreated(other, "Someone Else", "0000", 5.0)])
        pub = StatePublisher(store)
        acct = uuid.UUID(int=7)
        actor = GpuPersistentActor(bl, str(acct), store, pub, assigned_partition=3)
        assert actor.get_state() is None
        r = actor.process_message(CreateAccount(acct, "Jane Doe", "1234", 1000.0))
        assert r == ACKSuccess(BankAccount(acct, "Jane Doe", "1234", 1000.0))
        assert actor.process_message(CreditAccount(acct, 100.0)) == ACKSuccess(BankAccount(acct, "Jane Doe", "1234", 1100.0))
        n = len(pub.published)
        r = actor.process_message(DebitAccount(acct, 2000.0))
        assert isinstance(r, ACKError) and isinstance(r.exception, InsufficientFundsException) and len(pub.published) == n
        assert actor.process_message(DebitAccount(acct, 100.25)) == ACKSuccess(BankAccount(acct, "Jane Doe", "1234", 999.75))
        n = len(pub.published)
        # creating an existing account yields no events and the same state: nothing to publish (PersistentActor.scala:212)
        assert actor.process_message(CreateAccount(acct, "Mallory", "6666", 1.0)) == ACKSuccess(BankAccount(acct, "Jane Doe", "1234", 999.75))
        assert len(pub.published) == n
        stranger = GpuPersistentActor(bl, str(uuid.UUID(int=8)), store, pub)
        r = stranger.process_message(CreditAccount(uuid.UUID(int=8), 1.0))
        assert isinstance(r, ACKError) and isinstance(r.exception, AccountDoesNotExistException) and len(pub.published) == n
        assert [[type(x) for x in b] for b in pub.published] == [[StateRecord]] * 3  # create, credit, debit
        last = pub.published[-1][-1]
        assert (last.topic, last.partition, last.key) == ("bank-account-state", 3, str(acct))
        pub.ktable_progress()  # the account was born after recovery: the resident state grows, the three events fold onto it
        assert json.loads(store.get_aggregate_bytes(str(acct))) == json.loads(last.value)
        assert store.get_aggregate(str(acct)) == BankAccount(acct, "Jane Doe", "1234", 999.75)
        assert store.get_aggregate(str(other)) == BankAccount(other, "Someone Else", "0000", 5.0)
        assert GpuPersistentActor(bl, str(acct), store, pub).get_state() == BankAccount(acct, "Jane Doe", "1234", 999.75)
    finally:
        store.close()
