"""Host mirror of the user-facing command model (``surge.scaladsl.command``).

* ``AggregateCommandModel`` — ``modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/
  command/CommandModels.scala:12-31``: ``processCommand`` + ``handleEvent``; ``toCore`` folds
  ``events.foldLeft(state)(handleEvent)`` (:20, :26).  That fold, one aggregate and typically one
  event at a time, is the *semantic contract* (seam S3).
* ``SurgeCommandBusinessLogic`` — ``.../command/SurgeCommandBusinessLogic.scala:8-14`` and
  ``modules/command-engine/core/src/main/scala/surge/core/commondsl/SurgeGenericBusinessLogicTrait.scala:16-64``.
* ``BatchedAggregateCommandModel`` (``AsyncAggregateCommandModel``, :33-57) and ``AggregateEventModel``
  (``.../scaladsl/event/AggregateEventModel.scala:10-22``): the batched flavours that compile to the same core
  ``handle`` / ``applyAsync`` pair.  The Java DSL twin (``javadsl/.../CommandModels.scala:17-40``) is the
  same fold over ``java.util.Optional`` — DSL sugar (out of scope): not mirrored.
* ``ReplayableCommandModel`` — ADDITIVE.  ``handleEvent`` is arbitrary code and cannot run on a
  GPU: a model that wants GPU replay also declares its event algebra and the
  fixed-width encodings of its events and state.  ``tests/test_host_models.py`` checks that the
  declaration and the literal ``handle_event`` agree.
"""
from __future__ import annotations

from typing import Generic, Optional, Sequence, Tuple, TypeVar

import numpy as np

from .core import (
    KafkaTopic,
    SurgeAggregateReadFormatting,
    SurgeAggregateWriteFormatting,
    SurgeContext,
    SurgeEventWriteFormatting,
    SurgeProcessingModel,
)
from .kafka import KafkaPartitioner, PartitionStringUpToColon
from .schema import EVENT_DTYPE, EventAlgebra

Agg = TypeVar("Agg")
Cmd = TypeVar("Cmd")
Evt = TypeVar("Evt")


class AggregateCommandModel(Generic[Agg, Cmd, Evt]):
    def process_command(self, aggregate: Optional[Agg], command: Cmd) -> Sequence[Evt]:
        """``processCommand`` — returns the events, raises where the reference returns ``Failure``."""
        raise NotImplementedError

    def handle_event(self, aggregate: Optional[Agg], event: Evt) -> Optional[Agg]:
        raise NotImplementedError

    def to_core(self) -> SurgeProcessingModel:
        model = self

        class _Core(SurgeProcessingModel):
            def handle(self, ctx: SurgeContext, state, msg):
                events = list(model.process_command(state, msg))
                new_state = state
                for e in events:  # events.foldLeft(state)(handleEvent) — CommandModels.scala:20
                    new_state = model.handle_event(new_state, e)
                return ctx.persist_events(events).update_state(new_state).reply(lambda s: s)

            def apply_async(self, ctx: SurgeContext, state, events):
                new_state = state
                for e in events:  # CommandModels.scala:26
                    new_state = model.handle_event(new_state, e)
                return ctx.update_state(new_state).reply(lambda s: s)

        return _Core()


class BatchedAggregateCommandModel(Generic[Agg, Cmd, Evt]):
    """``AsyncAggregateCommandModel`` — ``CommandModels.scala:33-57``: the model supplies the BATCHED fold
    ``handleEvents(aggregate, Seq[Evt])`` itself (already the "whole segment at once" shape the GPU path has); ``toCore``
    calls it once per command / ``ApplyEvents`` (:41-43, :50-54).  Synchronous here."""

    def process_command(self, aggregate: Optional[Agg], command: Cmd) -> Sequence[Evt]:
        raise NotImplementedError

    def handle_events(self, aggregate: Optional[Agg], events: Sequence[Evt]) -> Optional[Agg]:
        raise NotImplementedError

    def to_core(self) -> SurgeProcessingModel:
        model = self

        class _Core(SurgeProcessingModel):
            def handle(self, ctx: SurgeContext, state, msg):
                events = list(model.process_command(state, msg))
                return ctx.persist_events(events).update_state(model.handle_events(state, events)).reply(lambda s: s)

            def apply_async(self, ctx: SurgeContext, state, events):
                return ctx.update_state(model.handle_events(state, list(events))).reply(lambda s: s)

        return _Core()


class AggregateEventModel(Generic[Agg, Evt]):
    """``surge.scaladsl.event.AggregateEventModel`` — ``modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/event/
    AggregateEventModel.scala:10-22``: an event-only aggregate; ``handleEvents(state, events)`` is the whole fold and
    commands are refused (:14-16)."""

    def handle_events(self, state: Optional[Agg], events: Sequence[Evt]) -> Optional[Agg]:
        raise NotImplementedError

    def to_core(self) -> SurgeProcessingModel:
        model = self

        class _Core(SurgeProcessingModel):
            def handle(self, ctx: SurgeContext, state, msg):
                raise NotImplementedError("Should not attempt to handle commands via AggregateEventModel")  # UnsupportedOperationException

            def apply_async(self, ctx: SurgeContext, state, events):
                return ctx.update_state(model.handle_events(state, list(events))).reply(lambda s: s)

        return _Core()


class ReplayableCommandModel(AggregateCommandModel[Agg, Cmd, Evt]):
    """An ``AggregateCommandModel`` that also declares how the GPU may replay it."""

    def event_algebra(self) -> EventAlgebra:
        raise NotImplementedError

    def encode_event(self, event: Evt) -> Tuple[int, int, Optional[int], Optional[float]]:
        """``(type, sequenceNumber, int payload or None, f64 payload or None)`` of one event."""
        raise NotImplementedError

    def aggregate_id_of(self, event: Evt) -> str:
        raise NotImplementedError

    def state_from_fixed(self, aggregate_id: str, fixed: np.void) -> Agg:
        """Rebuild the plugin's aggregate from the engine's 64-byte state (key re-attached here)."""
        raise NotImplementedError

    def state_to_fixed(self, aggregate: Agg) -> np.ndarray:
        raise NotImplementedError

    def event_json_template(self):
        """Optional: an ``EventJsonTemplate`` describing the JSON text ``SurgeEventWriteFormatting.write_event`` produces, so
        that an events topic can be decoded in the library (``surge_ingest_drain_json``) instead of record by record
        through ``read_event`` + ``encode_event``.  ``None`` = no template (the store falls back to the plugin's reader)."""
        return None

    def encode_events(self, events: Sequence[Evt]) -> np.ndarray:
        out = np.zeros(len(events), dtype=EVENT_DTYPE)
        for i, e in enumerate(events):
            ty, seq, arg, val = self.encode_event(e)
            out["type"][i] = ty
            out["seq"][i] = seq
            if val is not None:
                out["raw"][i] = np.float64(val).view(np.uint64)
            elif arg is not None:
                out["raw"][i] = np.uint64(np.int32(arg).astype(np.uint32))
        return out


class SurgeCommandBusinessLogic(Generic[Agg, Cmd, Evt]):
    """Plugin surface: names, topics, formats, model (SurgeGenericBusinessLogicTrait.scala:16-64)."""

    aggregate_name: str = ""
    state_topic: KafkaTopic = KafkaTopic("")
    events_topic: KafkaTopic = KafkaTopic("")
    publish_state_only: bool = False

    def command_model(self) -> AggregateCommandModel[Agg, Cmd, Evt]:
        raise NotImplementedError

    def aggregate_read_formatting(self) -> SurgeAggregateReadFormatting[Agg]:
        raise NotImplementedError

    def aggregate_write_formatting(self) -> SurgeAggregateWriteFormatting[Agg]:
        raise NotImplementedError

    def event_write_formatting(self) -> SurgeEventWriteFormatting[Evt]:
        raise NotImplementedError

    def partitioner(self) -> KafkaPartitioner:
        # default: PartitionStringUpToColon (SurgeGenericBusinessLogicTrait.scala:35)
        return PartitionStringUpToColon.instance
