"""TEST / BENCH INFRASTRUCTURE: synthetic events topics at scale (``tests/native/topic_gen.c``).

``counter_records`` writes the record keys and play-json values of Counter-fixture events in bulk (what
``CounterEventFormat.write_event`` writes, one event at a time, in ``examples/fixture_models.py``);
``frame_partitions`` turns them into Kafka record batches per partition with the product's own record-batch writer
(``surge_amd.snapshot.RecordBatchWriter`` — message format v2, batches closed at 16 KiB like the reference's producer,
``kafka.publisher.batch-size = 16384`` in ``reference.conf:115``, lz4 like its ``compression.type``, ``:112``).
``WireTopic`` frames them with the INDEPENDENT test-side writer instead (``tests/native/wire_writer.c``, the C twin of
``tests/kafka_wire.py``: no code shared with the product), in the layout the reference's publisher really produces — one
transaction per flush per partition, closed by a COMMIT marker, the odd flush aborted and retried."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "native", "topic_gen.c")
_LIB = os.path.join(_HERE, "native", "libtopic_gen.so")
_WSRC = os.path.join(_HERE, "native", "wire_writer.c")
_WLIB = os.path.join(_HERE, "native", "libwire_writer.so")
_lib = None
_wlib = None


def _build_one(src: str, lib: str, force: bool) -> str:
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        # partitions / records side by side where gcc has OpenMP (generation is outside every timed region: it only shortens the wait)
        if subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", src, "-o", lib + ".tmp"], capture_output=True).returncode != 0:
            subprocess.run(["gcc", "-O2", "-Wno-unknown-pragmas", "-shared", "-fPIC", src, "-o", lib + ".tmp"], check=True)
        os.replace(lib + ".tmp", lib)
    return lib


def build(force: bool = False) -> str:
    _build_one(_WSRC, _WLIB, force)
    return _build_one(_SRC, _LIB, force)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.surge_test_counter_records.restype = ctypes.c_int64
        _lib.surge_test_counter_records.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 8
        _lib.surge_test_bank_records.restype = ctypes.c_int64
        _lib.surge_test_bank_records.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 7
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


def bank_records(agg, seq, cents):
    """``(keys_utf8, key_off, values, val_off)`` for BankAccount events (tests/native/topic_gen.c): ``seq`` 1 = the account's
    ``BankAccountCreated`` with balance ``cents / 100``, above = a ``BankAccountUpdated`` with that new balance; the key is
    the account's UUID alone.  Views of reused buffers, like ``counter_records``."""
    lib = _load()
    agg, cents = np.ascontiguousarray(agg, dtype=np.int64), np.ascontiguousarray(cents, dtype=np.int64)
    seq = np.ascontiguousarray(seq, dtype=np.int32)
    n = agg.shape[0]
    if _bank_buffers.get("n", -1) < n:
        _bank_buffers.update(n=n, keys=np.empty(36 * n + 8, np.uint8), vals=np.empty(192 * n + 8, np.uint8), ko=np.empty(n + 1, np.int64), vo=np.empty(n + 1, np.int64))
    keys, vals, ko, vo = _bank_buffers["keys"], _bank_buffers["vals"], _bank_buffers["ko"][: n + 1], _bank_buffers["vo"][: n + 1]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.surge_test_bank_records(n, p(agg), p(seq), p(cents), p(keys), p(ko), p(vals), p(vo))
    return keys[: ko[n]], ko, vals[: vo[n]], vo


_bank_buffers = {}


def set_record_headers(headers=()):
    """Every record the independent writer writes from now on carries these ``(key: str, value: bytes)`` headers (none:
    the default).  Process-wide (a static of tests/native/wire_writer.c)."""
    def varint(v):
        z = (v << 1) ^ (v >> 63)
        out = bytearray()
        while z >= 0x80:
            out.append((z & 0x7F) | 0x80)
            z >>= 7
        out.append(z)
        return bytes(out)

    section = b""
    if headers:
        section = varint(len(headers))
        for k, v in headers:
            kb = k.encode("utf-8")
            section += varint(len(kb)) + kb + varint(len(v)) + v
    if wire_lib().surge_test_wire_set_record_headers(section, len(section)) != 0:
        raise ValueError("headers section longer than 255 bytes")
    return len(section)


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


def wire_lib():
    """``tests/native/libwire_writer.so`` with its signatures set."""
    global _wlib
    if _wlib is None:
        build()
        L = ctypes.CDLL(_WLIB)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        L.surge_test_wire_crc32c.restype, L.surge_test_wire_crc32c.argtypes = ctypes.c_uint32, [vp, i64]
        L.surge_test_wire_xxh32_short.restype, L.surge_test_wire_xxh32_short.argtypes = ctypes.c_uint32, [vp, i32, ctypes.c_uint32]
        L.surge_test_wire_lz4_frame.restype, L.surge_test_wire_lz4_frame.argtypes = i64, [vp, i64, vp]
        L.surge_test_wire_batch.restype, L.surge_test_wire_batch.argtypes = i64, [vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i64]
        L.surge_test_wire_control.restype, L.surge_test_wire_control.argtypes = i64, [vp, i64, i64, i32, i32, i64]
        L.surge_test_wire_topic_create.restype, L.surge_test_wire_topic_create.argtypes = vp, [i32]
        L.surge_test_wire_topic_destroy.restype, L.surge_test_wire_topic_destroy.argtypes = None, [vp]
        L.surge_test_wire_topic_partition.restype, L.surge_test_wire_topic_partition.argtypes = vp, [vp, i32, ctypes.POINTER(i64)]
        L.surge_test_wire_topic_end_offset.restype, L.surge_test_wire_topic_end_offset.argtypes = i64, [vp, i32]
        L.surge_test_wire_topic_fetch.restype, L.surge_test_wire_topic_fetch.argtypes = i32, [vp, i64, vp, vp, vp, vp, vp, i64, i64, i32, i64, i32, vp]
        L.surge_test_wire_set_record_headers.restype, L.surge_test_wire_set_record_headers.argtypes = i32, [ctypes.c_char_p, i64]
        _wlib = L
    return _wlib


WIRE_LZ4, WIRE_TRANSACTIONAL, WIRE_CONTROL = 1, 2, 4
COUNT_NAMES = ("data_batches", "control_batches", "records_written", "records_aborted", "transactions", "bytes")


class WireTopic:
    """An events topic over ``n_partitions`` partitions written fetch response by fetch response with the independent
    writer.  ``flush_events`` = K: partition p's records go out K per flush, every flush one transaction (data batches closed
    by the flush or at ``max_batch_bytes``, then a COMMIT marker; ``KafkaProducerActorImpl.scala:421-453``); every
    ``abort_every``-th flush first fails (the same records + an ABORT marker) and is retried; on partitions
    ``p % hold_markers == 1`` a fetch's last marker only arrives with the next fetch.  ``flush_events`` = 0: plain batches
    closed at ``max_batch_bytes`` only (the product writer's layout).  ``counts`` accumulates over the fetches."""

    def __init__(self, n_partitions: int, flush_events: int = 0, max_batch_bytes: int = 16384, codec: str = "lz4", abort_every: int = 0, hold_markers: int = 0):
        self._lib = wire_lib()
        self.n_partitions = n_partitions
        self._h = self._lib.surge_test_wire_topic_create(n_partitions)
        if not self._h:
            raise MemoryError("wire_writer")
        self._args = (int(flush_events), int(max_batch_bytes), 1 if codec == "lz4" else 0, int(abort_every), int(hold_markers))
        self._counts = np.zeros(8, np.int64)

    def close(self):
        if self._h:
            self._lib.surge_test_wire_topic_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def counts(self) -> dict:
        return dict(zip(COUNT_NAMES, (int(x) for x in self._counts)))

    def end_offsets(self):
        return [int(self._lib.surge_test_wire_topic_end_offset(self._h, p)) for p in range(self.n_partitions)]

    def fetch(self, partition, keys, key_off, values, val_off, last: bool = False):
        """The next fetch response: one ``bytes`` per partition (``None`` where a partition got nothing).  No records
        (``partition`` empty) = only the markers held back so far.  ``last``: nothing is held back at the end of this
        response (the topic ends here; the markers the response before held back still arrive with it)."""
        partition = np.ascontiguousarray(partition, dtype=np.int32)
        n = partition.shape[0]
        if n == 0:
            keys, values = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
            key_off, val_off = np.zeros(1, np.int64), np.zeros(1, np.int64)
        key_off, val_off = np.ascontiguousarray(key_off, dtype=np.int64), np.ascontiguousarray(val_off, dtype=np.int64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        args = self._args[:4] + (0,) if last else self._args
        rc = self._lib.surge_test_wire_topic_fetch(self._h, n, p(partition), p(keys), p(key_off), p(values), p(val_off), *args, p(self._counts))
        if rc != 0:
            raise MemoryError("wire_writer")
        out = []
        ln = ctypes.c_int64()
        for q in range(self.n_partitions):
            addr = self._lib.surge_test_wire_topic_partition(self._h, q, ctypes.byref(ln))
            out.append(ctypes.string_at(addr, ln.value) if ln.value else None)
        return out
