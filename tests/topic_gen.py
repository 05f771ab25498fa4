"""TEST / BENCH INFRASTRUCTURE: synthetic events topics at scale (``tests/native/topic_gen.c``).

``counter_records`` writes the record keys and play-json values of Counter-fixture events in bulk (what
``CounterEventFormat.write_event`` writes, one event at a time, in ``examples/fixture_models.py``);
``frame_partitions`` turns them into Kafka record batches per partition with the product's own record-batch writer
(``surge_amd.snapshot.RecordBatchWriter`` — message format v2, batches closed at 16 KiB like the reference's producer,
``kafka.publisher.batch-size = 16384`` in ``reference.conf:115``, lz4 like its ``compression.type``, ``:112``)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "native", "topic_gen.c")
_LIB = os.path.join(_HERE, "native", "libtopic_gen.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", _SRC, "-o", _LIB + ".tmp"], check=True)
        os.replace(_LIB + ".tmp", _LIB)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.surge_test_counter_records.restype = ctypes.c_int64
        _lib.surge_test_counter_records.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 8
    return _lib


_buffers = {}


def counter_records(agg, ev_type, arg, seq):
    """``(keys_utf8, key_off, values, val_off)`` for Counter events: ``ev_type`` 0 increment / 1 decrement / 2 no-op.
    The arrays are views of buffers this module keeps and reuses: valid until the next call."""
    lib = _load()
    agg = np.ascontiguousarray(agg, dtype=np.int64)
    n = agg.shape[0]
    ev_type, arg, seq = (np.ascontiguousarray(a, dtype=np.int32) for a in (ev_type, arg, seq))
    if _buffers.get("n", -1) < n:  # (fresh pages fault in at ~0.5 us per record: keep them)
        _buffers.update(n=n, keys=np.empty(40 * n + 8, np.uint8), vals=np.empty(128 * n + 8, np.uint8), ko=np.empty(n + 1, np.int64), vo=np.empty(n + 1, np.int64))
    keys, vals, ko, vo = _buffers["keys"], _buffers["vals"], _buffers["ko"][: n + 1], _buffers["vo"][: n + 1]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.surge_test_counter_records(n, p(agg), p(ev_type), p(arg), p(seq), p(keys), p(ko), p(vals), p(vo))
    return keys[: ko[n]], ko, vals[: vo[n]], vo


def frame_partitions(writer, partition, keys, key_off, values, val_off, timestamp_ms: int = 0):
    """The records as record batches, one ``bytes`` per partition (``None`` where a partition got nothing), through
    ``writer`` (a ``RecordBatchWriter``; its per-partition offsets continue from call to call)."""
    writer.reset()
    writer.append(None, partition, keys, key_off, values, val_off, timestamp_ms)
    out = []
    for p in range(writer.n_partitions):
        data, nrec, _ = writer.partition_bytes(p)
        out.append(data if nrec else None)
    return out
