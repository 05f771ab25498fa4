"""Shard map: ``aggregateId -> partition`` exactly as the reference routes commands and records.

``partitionForKey = abs(MurmurHash3.stringHash(partitionByString) % numberOfPartitions)`` —
``modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8``; the default partitioner cuts
the key at the first ``':'`` (``PartitionStringUpToColon``, :38-42) so events keyed
``"<id>:<seq>"`` land with their aggregate.  The hash itself is computed by the C ABI
(``surge_replay_partition_hash[_up_to_colon]``: CPU; ``..._device``: GPU kernel K4); the cut belongs to the
partitioner's ``partitionBy``, not to ``partitionForKey``.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from . import _native


def utf16_table(keys: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate strings as UTF-16 code units (what JVM ``String.charAt`` sees) + offsets."""
    parts = [np.frombuffer(k.encode("utf-16-le"), dtype=np.uint16) for k in keys]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    if parts:
        np.cumsum([p.size for p in parts], out=off[1:])
    data = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint16)
    return np.ascontiguousarray(data, dtype=np.uint16), off


def partition_for_keys(keys: Sequence[str], n_partitions: int, up_to_colon: bool = False) -> np.ndarray:
    """Batch ``partitionForKey`` (CPU entry point): hashes each WHOLE string, like ``KafkaPartitioner.scala:8``;
    ``up_to_colon=True`` first applies ``PartitionStringUpToColon.partitionBy`` (``:38-42``)."""
    if n_partitions <= 0:
        raise ValueError("numberOfPartitions must be positive")
    data, off = utf16_table(keys)
    out = np.zeros(len(keys), dtype=np.int32)
    if len(keys) == 0:
        return out
    lib = _native.load()
    fn = lib.surge_replay_partition_hash_up_to_colon if up_to_colon else lib.surge_replay_partition_hash
    rc = fn(
        data.ctypes.data_as(ctypes.c_void_p) if data.size else None,
        off.ctypes.data_as(ctypes.c_void_p),
        len(keys),
        n_partitions,
        out.ctypes.data_as(ctypes.c_void_p),
    )
    if rc != 0:
        raise RuntimeError(f"surge_replay_partition_hash failed: {rc}")
    return out


class KafkaPartitionProvider:
    """``trait KafkaPartitionProvider`` (KafkaPartitioner.scala:7-9): hashes the string it is given, whole."""

    def partition_for_key(self, partition_by_string: str, number_of_partitions: int) -> int:
        return int(partition_for_keys([partition_by_string], number_of_partitions)[0])

    def partition_for(self, key: str, number_of_partitions: int) -> int:
        """What the producer / router computes for a record key: ``partitionForKey(partitionBy(key), n)``
        (``KafkaProducer.scala:45-57``)."""
        by = getattr(self, "partition_by", None)
        return self.partition_for_key(by(key) if by is not None else key, number_of_partitions)


class KafkaPartitioner(KafkaPartitionProvider):
    """``KafkaPartitioner[Key]`` (KafkaPartitioner.scala:21-24)."""

    def partition_by(self, key: str) -> str:
        raise NotImplementedError

    @property
    def optional_partition_by(self) -> Optional[Callable[[str], str]]:
        return self.partition_by


class StringIdentityPartitioner(KafkaPartitioner):
    """Hashes the whole key, colons included (KafkaPartitioner.scala:29-31)."""

    instance: "StringIdentityPartitioner"

    def partition_by(self, key: str) -> str:
        return key


class PartitionStringUpToColon(KafkaPartitioner):
    """Default partitioner (KafkaPartitioner.scala:38-42)."""

    instance: "PartitionStringUpToColon"

    def partition_by(self, key: str) -> str:
        return key.split(":", 1)[0]


class NoPartitioner(KafkaPartitionProvider):
    """``NoPartitioner`` (KafkaPartitioner.scala:17-19): let Kafka pick; no partitionBy."""

    optional_partition_by = None


StringIdentityPartitioner.instance = StringIdentityPartitioner()
PartitionStringUpToColon.instance = PartitionStringUpToColon()
