"""Shard map: ``aggregateId -> partition`` exactly as the reference routes commands and records.

``partitionForKey = abs(MurmurHash3.stringHash(partitionByString) % numberOfPartitions)`` —
``modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8``; the default partitioner cuts
the key at the first ``':'`` (``PartitionStringUpToColon``, :38-42) so events keyed
``"<id>:<seq>"`` land with their aggregate.  The hash itself is computed by the C ABI
(``surge_replay_partition_hash``: CPU; ``surge_replay_partition_hash_device``: GPU kernel K4).
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


def partition_for_keys(keys: Sequence[str], n_partitions: int) -> np.ndarray:
    """Batch ``partitionForKey`` with ``PartitionStringUpToColon`` semantics (CPU entry point)."""
    if n_partitions <= 0:
        raise ValueError("numberOfPartitions must be positive")
    data, off = utf16_table(keys)
    out = np.zeros(len(keys), dtype=np.int32)
    if len(keys) == 0:
        return out
    lib = _native.load()
    rc = lib.surge_replay_partition_hash(
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
    def partition_for_key(self, partition_by_string: str, number_of_partitions: int) -> int:
        # the C entry point already applies takeWhile(_ != ':'); a key without ':' hashes whole
        return int(partition_for_keys([partition_by_string], number_of_partitions)[0])


class KafkaPartitioner(KafkaPartitionProvider):
    """``KafkaPartitioner[Key]`` (KafkaPartitioner.scala:21-24)."""

    def partition_by(self, key: str) -> str:
        raise NotImplementedError

    @property
    def optional_partition_by(self) -> Optional[Callable[[str], str]]:
        return self.partition_by


class StringIdentityPartitioner(KafkaPartitioner):
    """Hashes the whole key, colons included (KafkaPartitioner.scala:30-32).

    The C entry points restate the default routing (cut at ``':'`` then hash), so this rarely
    used partitioner hashes on the host instead.
    """

    instance: "StringIdentityPartitioner"

    def partition_by(self, key: str) -> str:
        return key

    def partition_for_key(self, partition_by_string: str, number_of_partitions: int) -> int:
        return _partition_whole(partition_by_string, number_of_partitions)


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


def _partition_whole(s: str, n: int) -> int:
    """Hash a full string including any ':' (host-only path for StringIdentityPartitioner)."""
    u = np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16).astype(np.uint32)
    M = 0xFFFFFFFF

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & M

    def mix_last(h, k):
        k = (k * 0xCC9E2D51) & M
        k = rotl(k, 15)
        k = (k * 0x1B873593) & M
        return h ^ k

    h = 0xF7CA7FD2
    i = 0
    while i + 1 < len(u):
        h = mix_last(h, ((int(u[i]) << 16) + int(u[i + 1])) & M)
        h = rotl(h, 13)
        h = (h * 5 + 0xE6546B64) & M
        i += 2
    if i < len(u):
        h = mix_last(h, int(u[i]))
    h ^= len(u)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    h ^= h >> 16
    signed = h - (1 << 32) if h & 0x80000000 else h
    r = abs(signed) % n if signed >= 0 else -((-signed) % n)  # JVM truncated %
    return abs(r)
