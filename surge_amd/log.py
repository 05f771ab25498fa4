"""CSR packing of an events topic: group records by aggregate, preserving Kafka offset order.

The step immediately before the fold (host-side part).  Input is what a consumer
of the events topic sees: records in offset order whose keys are ``"<aggregateId>:<seq>"``
(``modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:122-124``).
Ordering is by arrival (offset), never by ``seq``: a ``NoOpEvent`` consumes a sequence number without
storing it, so keys can repeat.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .command import ReplayableCommandModel
from .schema import EVENT_DTYPE


@dataclass
class KeyTable:
    """Dense index <-> aggregate id.  Strings never go to the GPU."""

    keys: List[str] = field(default_factory=list)
    index: Dict[str, int] = field(default_factory=dict)

    def intern(self, key: str) -> int:
        i = self.index.get(key)
        if i is None:
            i = len(self.keys)
            self.keys.append(key)
            self.index[key] = i
        return i

    def get(self, key: str) -> Optional[int]:
        return self.index.get(key)

    def __len__(self) -> int:
        return len(self.keys)


@dataclass
class EventLog:
    seg_off: np.ndarray   # int64[A+1]
    events: np.ndarray    # EVENT_DTYPE[E]
    keys: KeyTable

    @property
    def n_aggregates(self) -> int:
        return self.seg_off.shape[0] - 1


def group_by_aggregate(agg_idx: np.ndarray, events: np.ndarray, n_agg: int) -> Tuple[np.ndarray, np.ndarray]:
    """Stable group-by: returns ``(seg_off, events_sorted)`` with arrival order kept inside a group."""
    agg_idx = np.asarray(agg_idx, dtype=np.int64)
    order = np.argsort(agg_idx, kind="stable")
    counts = np.bincount(agg_idx, minlength=n_agg).astype(np.int64)
    seg_off = np.zeros(n_agg + 1, dtype=np.int64)
    np.cumsum(counts, out=seg_off[1:])
    return seg_off, np.ascontiguousarray(events[order])


def pack_events(model: ReplayableCommandModel, events_in_offset_order: Sequence, keys: Optional[KeyTable] = None,
                capacity: int = 0) -> EventLog:
    """Encode domain events with the model's fixed-width codec and CSR-pack them.

    ``capacity`` reserves extra (empty) aggregates so later micro-batches may introduce new ids.
    """
    keys = keys if keys is not None else KeyTable()
    enc = model.encode_events(events_in_offset_order)
    agg_idx = np.fromiter((keys.intern(model.aggregate_id_of(e)) for e in events_in_offset_order), dtype=np.int64,
                          count=len(events_in_offset_order))
    n_agg = max(len(keys), capacity)
    seg_off, ev = group_by_aggregate(agg_idx, enc, n_agg)
    return EventLog(seg_off, ev, keys)


def pack_batch(model: ReplayableCommandModel, events_in_offset_order: Sequence, keys: KeyTable, n_agg: int):
    """Micro-batch for ``append_fold``: ``(group_agg, group_off, events)``, one group per touched aggregate."""
    enc = model.encode_events(events_in_offset_order)
    ids = [model.aggregate_id_of(e) for e in events_in_offset_order]
    # check the capacity BEFORE interning anything: a rejected batch must not leave ids in the key table
    fresh = list(dict.fromkeys(k for k in ids if keys.get(k) is None))
    if len(keys) + len(fresh) > n_agg:
        raise IndexError(f"aggregate {fresh[max(0, n_agg - len(keys))]!r} exceeds the store capacity {n_agg} "
                         "(grow the resident state first: ReplayEngine.grow)")
    agg_idx = np.fromiter((keys.intern(k) for k in ids), dtype=np.int64, count=len(ids))
    return batch_groups(agg_idx, enc)


def batch_groups(agg_idx: np.ndarray, events: np.ndarray):
    """Group an already-encoded micro-batch: unique aggregates (ascending), offsets, sorted events."""
    agg_idx = np.asarray(agg_idx, dtype=np.int64)
    order = np.argsort(agg_idx, kind="stable")
    sorted_idx = agg_idx[order]
    group_agg, starts = np.unique(sorted_idx, return_index=True)
    group_off = np.concatenate([starts.astype(np.int64), np.array([agg_idx.shape[0]], dtype=np.int64)])
    return group_agg.astype(np.int64), group_off, np.ascontiguousarray(events[order], dtype=EVENT_DTYPE)
