"""Host mirror of the CALLERS of the fold on top of the GPU-recovered store.

The reference folds inside one actor per aggregate; what surrounds the fold there decides WHAT is published and WHEN an
actor may trust the store.  This module restates exactly that protocol — synchronously, no actors — so the GPU store can
be checked end to end against the literal expectations of the reference's own specs:

* ``PersistentActor.handle`` / ``processMessage`` / ``doApplyEvent`` / ``callEventHandler`` —
  ``modules/command-engine/core/src/main/scala/surge/internal/persistence/PersistentActor.scala:197-272``:
  a command publishes its events AND the new state in one batch iff ``events.nonEmpty || records.nonEmpty ||
  state changed`` (:212); ``ApplyEvents`` publishes ONLY the state and only ``state changed`` (:255-257); an exception in
  ``processCommand`` / ``handleEvent`` becomes ``ACKError`` and leaves the actor's state untouched (:227-229, :260-262);
  ``publishStateOnly`` drops the event records (:207-211).
* ``SurgeModel.serializeState`` / ``serializeEvents`` — ``.../internal/SurgeModel.scala:37-65``: the state record is
  ``(stateTopic, assignedPartition, key = aggregateId, value = writeState(s).value | null, headers)``.
* ``KTableInitializationSupport.initializeState`` / ``fetchState`` — ``.../KTableInitializationSupport.scala:37-81``:
  ask the producer ``isAggregateStateCurrent``; not current -> retry after ``initialize-state-interval`` (500 ms);
  current -> ``getAggregateBytes`` + ``deserializeState``; a failed read retries after ``fetch-state-retry-interval``
  (2 s); more than ``max-initialization-attempts`` (10) -> the actor fails with ``AggregateInitializationException``
  (``reference.conf`` of common :137-142, ``PersistentActor.scala:328-333``).
* ``KafkaProducerActorState.inFlight`` / ``processedUpTo`` / ``IsAggregateStateCurrent`` —
  ``.../internal/kafka/KafkaProducerActorImpl.scala:530-540, 684-705``: a published state record is "in flight" until
  the KTable has indexed its offset; an aggregate is current iff none of its records are in flight.

The state of record lives on the GPU: every published state change is folded onto the resident state with the K3
micro-batch path, and actors initialise from ``GpuReplayStateStore.get_aggregate_bytes`` (seam S2).  The literal
``handle_event`` of the model stays the semantic contract (seam S3) — the tests hold the two to the same bytes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Generic, List, Optional, Sequence, Tuple, TypeVar

from .command import SurgeCommandBusinessLogic
from .core import SurgeContext
from .snapshot import StateRecord
from .store import AggregateInitializationException, GpuReplayStateStore

Agg = TypeVar("Agg")


@dataclass(frozen=True)
class ACKSuccess(Generic[Agg]):
    """``PersistentActor.ACKSuccess(aggregateState: Option[S])`` (PersistentActor.scala:45-47)."""

    aggregate_state: Optional[Agg]


@dataclass(frozen=True)
class ACKError:
    """``PersistentActor.ACKError(exception)`` (:48-50)."""

    exception: BaseException


@dataclass(frozen=True)
class ACKRejection:
    """``PersistentActor.ACKRejection(rejection)`` (:60-62): the command was rejected by the model (``ctx.reject``) —
    distinct from ``ACKError``, nothing is published, the state is untouched."""

    rejection: object


@dataclass(frozen=True)
class EventRecord:
    topic: str
    key: str
    value: bytes
    headers: Dict[str, str] = field(default_factory=dict)


class AggregateStateNotCurrentInKTableException(RuntimeError):
    """``KTableInitializationSupport.scala:20-23``."""


@dataclass
class RetryConfig:
    """``surge.aggregate-actor`` retry settings (``reference.conf`` of common :137-142)."""

    initialize_state_interval_s: float = 0.5
    fetch_state_retry_interval_s: float = 2.0
    max_initialization_attempts: int = 10


class InFlightTracker:
    """The producer's view of "has the KTable caught up with what I published" for ONE state-topic partition
    (``KafkaProducerActorState``, KafkaProducerActorImpl.scala:660-705): the newest in-flight offset per key; a KTable
    progress update up to offset ``o`` retires every record with offset <= ``o``."""

    def __init__(self):
        self._in_flight: Dict[str, int] = {}

    def add_in_flight(self, records: Sequence[Tuple[str, int]]) -> None:
        """``(key, offset)`` of just-published state records; only the max offset per key is kept (:692-705)."""
        for key, offset in records:
            if offset > self._in_flight.get(key, -1):
                self._in_flight[key] = offset

    def processed_up_to(self, ktable_current_offset: int) -> None:
        """``KTableProgressUpdate(LagInfo(currentOffsetPosition = o, ...))`` (:684-698)."""
        self._in_flight = {k: o for k, o in self._in_flight.items() if o > ktable_current_offset}

    def in_flight_for_aggregate(self, aggregate_id: str) -> List[int]:
        return [self._in_flight[aggregate_id]] if aggregate_id in self._in_flight else []

    def is_aggregate_state_current(self, aggregate_id: str) -> bool:
        """``IsAggregateStateCurrent`` -> ``noRecordsInFlight`` (:530-534)."""
        return not self.in_flight_for_aggregate(aggregate_id)


class StatePublisher:
    """Where a batch of records goes (``KafkaProducerActor.publish``): appends to an in-memory "topic", assigns state-topic
    offsets, feeds the in-flight tracker, and folds the batch's EVENTS onto the GPU-resident state (K3) — the moment the
    reference's KTable would index the state record, ``ktable_progress`` marks it processed."""

    def __init__(self, store: GpuReplayStateStore, tracker: Optional[InFlightTracker] = None):
        self.store = store
        self.tracker = tracker or InFlightTracker()
        self.published: List[List[object]] = []  # one list per publish call (= one Kafka transaction)
        self._next_offset = 0
        self._pending_events: List[object] = []

    def publish(self, aggregate_id: str, records: Sequence[object], events: Sequence[object]) -> None:
        self.published.append(list(records))
        flight = []
        for r in records:
            if isinstance(r, StateRecord):
                flight.append((r.key, self._next_offset))
                self._next_offset += 1
        self.tracker.add_in_flight(flight)
        self._pending_events.extend(events)

    def ktable_progress(self) -> None:
        """The KTable catches up: pending events are folded onto the GPU store (one micro-batch for all aggregates) and
        every published offset is reported processed."""
        if self._pending_events:
            self.store.apply_events(self._pending_events)
            self._pending_events = []
        self.tracker.processed_up_to(self._next_offset - 1)


class GpuPersistentActor(Generic[Agg]):
    """One aggregate's ``PersistentActor``, synchronous.  ``sleep`` is injectable so tests do not wait."""

    def __init__(self, business_logic: SurgeCommandBusinessLogic, aggregate_id: str, store: GpuReplayStateStore,
                 publisher: StatePublisher, assigned_partition: int = 0, retry: Optional[RetryConfig] = None,
                 sleep: Callable[[float], None] = lambda s: None):
        self.business_logic = business_logic
        self.model = business_logic.command_model()
        self.core = self.model.to_core()
        self.aggregate_id = aggregate_id
        self.store = store
        self.publisher = publisher
        self.assigned_partition = assigned_partition
        self.retry = retry or RetryConfig()
        self.sleep = sleep
        self.state: Optional[Agg] = None
        self.initialized = False
        self.initialization_attempts = 0

    # -- KTableInitializationSupport ------------------------------------------------------------------------------
    def initialize(self) -> None:
        cause: Optional[BaseException] = None
        attempts = 0
        while True:
            if attempts > self.retry.max_initialization_attempts:
                self.initialization_attempts = attempts
                raise AggregateInitializationException(
                    f"Aggregate {self.aggregate_id} could not be initialized") from cause
            if not self.publisher.tracker.is_aggregate_state_current(self.aggregate_id):
                cause = AggregateStateNotCurrentInKTableException(self.aggregate_id)
                self.sleep(self.retry.initialize_state_interval_s)
                attempts += 1
                continue
            try:
                data = self.store.get_aggregate_bytes(self.aggregate_id)  # seam S2
                self.state = None if data is None else self.business_logic.aggregate_read_formatting().read_state(data)
                self.initialized = True
                self.initialization_attempts = attempts
                return
            except Exception as exc:  # failed read -> fetchState's recover -> retry
                cause = exc
                self.sleep(self.retry.fetch_state_retry_interval_s)
                attempts += 1

    # -- serialization (SurgeModel.scala:37-65) -----------------------------------------------------------------------
    def _serialize_state(self, state: Optional[Agg]) -> StateRecord:
        topic = self.business_logic.state_topic.name
        if state is None:
            return StateRecord(topic, self.assigned_partition, self.aggregate_id, None)
        ser = self.business_logic.aggregate_write_formatting().write_state(state)
        return StateRecord(topic, self.assigned_partition, self.aggregate_id, ser.value, dict(ser.headers))

    def _serialize_events(self, events: Sequence[Tuple[object, object]]) -> List[EventRecord]:
        fmt = self.business_logic.event_write_formatting()
        out = []
        for evt, topic in events:
            msg = fmt.write_event(evt)
            out.append(EventRecord((topic or self.business_logic.events_topic).name, msg.key, msg.value, dict(msg.headers)))
        return out

    # -- PersistentActor.handle (:197-232) -----------------------------------------------------------------------------
    def process_message(self, message) -> object:
        if not self.initialized:
            self.initialize()
        try:
            ctx = self.core.handle(SurgeContext(state=self.state, default_event_topic=self.business_logic.events_topic), self.state, message)
            if ctx.is_rejected:
                return ACKRejection(ctx.rejection)
            events = [e for e, _ in ctx.events]
            is_something_new = bool(events) or bool(ctx.records) or (self.state != ctx.state)
            records: List[object] = [] if self.business_logic.publish_state_only else list(self._serialize_events(ctx.events))
            records += list(ctx.records)
            records.append(self._serialize_state(ctx.state))
            if is_something_new:
                self.publisher.publish(self.aggregate_id, records, events)
            self.state = ctx.state
            return ACKSuccess(ctx.state)
        except Exception as exc:  # .recover { case e => ACKError(e) } — the actor's state is untouched
            return ACKError(exc)

    # -- PersistentActor.doApplyEvent (:245-264) -------------------------------------------------------------------------
    def apply_events(self, events: Sequence[object]) -> object:
        if not self.initialized:
            self.initialize()
        try:
            ctx = self.core.apply_async(SurgeContext(state=self.state, default_event_topic=self.business_logic.events_topic), self.state, list(events))
            if self.state != ctx.state:  # state-only publish, and only when it changed (:255-257)
                self.publisher.publish(self.aggregate_id, [self._serialize_state(ctx.state)], list(events))
            self.state = ctx.state
            return ACKSuccess(ctx.state)
        except Exception as exc:
            return ACKError(exc)

    def get_state(self) -> Optional[Agg]:
        """``PersistentActor.GetState`` -> ``StateResponse`` (:52-53)."""
        if not self.initialized:
            self.initialize()
        return self.state
