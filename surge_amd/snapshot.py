"""State-topic snapshot records from the GPU-resident store (host side).

The step immediately after the fold: make the recovered / updated state consumable by an unmodified
Surge node through its normal KTable.  One record per aggregate, exactly what
``SurgeModel.serializeState`` builds (``modules/command-engine/core/src/main/scala/surge/internal/SurgeModel.scala:57-65``):

    ProducerRecord(stateTopic, assignedPartition, key = aggregateId,
                   value = aggregateWriteFormatting.writeState(state).value  | null when the state is None,
                   headers = writeState(state).headers)

The state topic is log-compacted (``cleanup.policy=compact``,
``.../test/scala/surge/internal/domain/SurgeMessagePipelineSpec.scala:128-136``): ``compact`` below is that
last-record-per-key rule, which is also what the KTable applies on restore
(``SurgeStateStoreConsumer.scala:69``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from .kafka import partition_for_keys
from .schema import STATE_POISONED, STATE_PRESENT
from .store import GpuReplayStateStore


@dataclass(frozen=True)
class StateRecord:
    topic: str
    partition: int
    key: str
    value: Optional[bytes]  # None = tombstone
    headers: Dict[str, str] = field(default_factory=dict)


class SnapshotWriter:
    """Emits state-topic records for all, or only the touched, aggregates of a ``GpuReplayStateStore``."""

    def __init__(self, store: GpuReplayStateStore, n_partitions: int):
        self.store = store
        self.n_partitions = n_partitions
        self.topic = store.business_logic.state_topic.name

    def records_for(self, aggregate_ids: Sequence[str]) -> List[StateRecord]:
        n_agg = self.store.engine.n_agg  # ids interned beyond the resident state have no state yet (like get_aggregate)
        ids = [k for k in aggregate_ids if self.store.keys.get(k) is not None and self.store.keys.index[k] < n_agg]
        if not ids:
            return []
        idx = np.array([self.store.keys.index[k] for k in ids], dtype=np.int64)
        states = self.store.engine.gather(idx)
        parts = partition_for_keys(ids, self.n_partitions, up_to_colon=True)  # events and state share the default partitioner (KafkaPartitioner.scala:8,38-42)
        fmt = self.store.business_logic.aggregate_write_formatting()
        out = []
        for k, st, part in zip(ids, states, parts):
            fl = int(st["flags"])
            if fl & STATE_POISONED:
                continue  # replay of this aggregate failed: publish nothing rather than a state the JVM fold never had
            if fl & STATE_PRESENT:
                ser = fmt.write_state(self.store.model.state_from_fixed(k, st))
                out.append(StateRecord(self.topic, int(part), k, ser.value, dict(ser.headers)))
            else:
                out.append(StateRecord(self.topic, int(part), k, None))
        return out

    def full_snapshot(self) -> List[StateRecord]:
        return self.records_for(list(self.store.keys.keys))


def compact(records: Iterable[StateRecord]) -> Dict[str, Optional[bytes]]:
    """Kafka log compaction / KTable materialisation: last record per key wins; tombstones delete."""
    table: Dict[str, Optional[bytes]] = {}
    for r in records:
        if r.value is None:
            table.pop(r.key, None)
        else:
            table[r.key] = r.value
    return table
