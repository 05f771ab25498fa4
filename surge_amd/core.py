"""Host mirror of the reference's serialization surface and of the core model interface.

Names and argument meaning follow the reference so a Surge user finds the same contracts:

* ``SerializedAggregate`` / ``SerializedMessage`` —
  ``modules/serialization/src/main/scala/surge/core/SerializedAggregate.scala:7``,
  ``.../SerializedMessage.scala:6``
* ``SurgeAggregateReadFormatting.read_state`` / ``SurgeAggregateWriteFormatting.write_state`` /
  ``SurgeEventWriteFormatting.write_event`` — ``.../SurgeFormatting.scala:5-17``
* ``SurgeEventReadFormatting.read_event`` — ADDITIVE: the reference has no event reader
  (SURVEY §0.3); a replay engine needs one next to the plugin surface.
* ``SurgeContext`` / ``SurgeProcessingModel`` —
  ``modules/command-engine/core/src/main/scala/surge/internal/domain/AggregateProcessingModel.scala:17-64``
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Any, Callable, Dict, Generic, Optional, Sequence, Tuple, TypeVar

State = TypeVar("State")
Event = TypeVar("Event")
Message = TypeVar("Message")


@dataclass(frozen=True)
class SerializedAggregate:
    value: bytes
    headers: Dict[str, str] = field(default_factory=dict)


@dataclass(frozen=True)
class SerializedMessage:
    key: str
    value: bytes
    headers: Dict[str, str] = field(default_factory=dict)


class SurgeAggregateReadFormatting(Generic[State]):
    def read_state(self, data: bytes) -> Optional[State]:
        raise NotImplementedError


class SurgeAggregateWriteFormatting(Generic[State]):
    def write_state(self, state: State) -> SerializedAggregate:
        raise NotImplementedError


class SurgeAggregateFormatting(SurgeAggregateReadFormatting[State], SurgeAggregateWriteFormatting[State]):
    pass


class SurgeEventWriteFormatting(Generic[Event]):
    def write_event(self, evt: Event) -> SerializedMessage:
        raise NotImplementedError


class SurgeEventReadFormatting(Generic[Event]):
    """Additive: decode one record of the events topic (``key``, ``value``) back into an event."""

    def read_event(self, msg: SerializedMessage) -> Event:
        raise NotImplementedError


@dataclass(frozen=True)
class KafkaTopic:
    """``modules/common/src/main/scala/surge/kafka/KafkaTopic.scala`` (name only matters here)."""

    name: str


@dataclass(frozen=True)
class SurgeContext(Generic[State, Event]):
    """Immutable result carrier — ``SurgeContextImpl`` (AggregateProcessingModel.scala:36-64).

    ``replies`` stands in for the reply side effects (there is no actor to reply to here): each
    entry is the function applied to the final state.
    """

    state: Optional[State] = None
    default_event_topic: Optional[KafkaTopic] = None
    events: Tuple[Tuple[Any, Optional[KafkaTopic]], ...] = ()
    records: Tuple[Any, ...] = ()
    replies: Tuple[Callable[[Optional[State]], Any], ...] = ()
    is_rejected: bool = False
    rejection: Any = None

    def persist_event(self, event) -> "SurgeContext":
        return replace(self, events=self.events + ((event, self.default_event_topic),))

    def persist_events(self, events: Sequence[Any]) -> "SurgeContext":
        return replace(self, events=self.events + tuple((e, self.default_event_topic) for e in events))

    def persist_to_topic(self, event, topic: KafkaTopic) -> "SurgeContext":
        return replace(self, events=self.events + ((event, topic),))

    def persist_record(self, record) -> "SurgeContext":
        return replace(self, records=self.records + (record,))

    def update_state(self, state: Optional[State]) -> "SurgeContext":
        return replace(self, state=state)

    def reply(self, reply_with_message: Callable[[Optional[State]], Any]) -> "SurgeContext":
        return replace(self, replies=self.replies + (reply_with_message,))

    def reject(self, rejection) -> "SurgeContext":
        return replace(self, is_rejected=True, rejection=rejection)


class SurgeProcessingModel(Generic[State, Message, Event]):
    """``handle`` / ``applyAsync`` (AggregateProcessingModel.scala:17-22); synchronous here."""

    def handle(self, ctx: SurgeContext, state: Optional[State], msg: Message) -> SurgeContext:
        raise NotImplementedError

    def apply_async(self, ctx: SurgeContext, state: Optional[State], events: Sequence[Event]) -> SurgeContext:
        raise NotImplementedError
