"""Events-topic ingest: Kafka record batches of one partition -> records in offset order
with their aggregate index, through the C ABI in ``include/surge_ingest.h`` (host side, no GPU).

``EventsTopicIngest.feed(bytes)`` takes what a fetch of the events topic returns (RecordBatch v2, lz4 or
uncompressed, ``read_committed`` by default like ``SurgeStateStoreConsumer.scala:38``);
``drain_fixed16()`` returns ``(agg_idx, events, offsets)`` for GPU-ready topics whose record value is the
16-byte fixed event, ``drain_records()`` returns ``(offset, key, value)`` tuples for plugin-decoded values.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _native
from .log import KeyTable
from .schema import EVENT_DTYPE

READ_UNCOMMITTED, READ_COMMITTED = 0, 1
FRAMES = 0x100  # SURGE_INGEST_FRAMES: frame on the host, decode the records on the GPU (DeviceDecoder)
DEVICE_LZ4 = 0x200  # SURGE_INGEST_DEVICE_LZ4: ... and leave LZ4 frames for the GPU as well
DEVICE_CRC = 0x400  # SURGE_INGEST_DEVICE_CRC: ... and finish a data batch's CRC-32C on the GPU (the host checksums its 40 header bytes only)
SECTION_CRC_PENDING = 0x100  # in a section's codec

SECTION_DTYPE = np.dtype([("byte_off", "<i8"), ("byte_len", "<i8"), ("base_offset", "<i8"), ("n_records", "<i4"), ("codec", "<i4")])
assert SECTION_DTYPE.itemsize == 32

RECORD_DTYPE = np.dtype([("offset", "<i8"), ("agg_idx", "<i8"), ("key_off", "<i8"), ("key_len", "<i4"),
                         ("value_len", "<i4"), ("value_off", "<i8")])
assert RECORD_DTYPE.itemsize == 40


EVJ_NAME, EVJ_MAX_TYPES = 64, 16
ARG_NONE, ARG_I32, ARG_F64 = 0, 1, 2


class _CEventJsonType(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * EVJ_NAME), ("event_type", ctypes.c_uint32), ("arg_kind", ctypes.c_uint32),
                ("seq_field", ctypes.c_char * EVJ_NAME), ("arg_field", ctypes.c_char * EVJ_NAME)]


class CEventJsonTemplate(ctypes.Structure):
    _fields_ = [("n_types", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("discriminator", ctypes.c_char * EVJ_NAME),
                ("types", _CEventJsonType * EVJ_MAX_TYPES)]


class EventJsonTemplate:
    """``surge_event_json_template``: how a model's JSON event text maps onto the 16-byte fixed event.

    ``types``: ``(discriminator value, event type, seq field or "", arg field or "", ARG_*)`` per event class;
    ``discriminator``: the field naming the class (play-json's sealed-family ``"_type"``), ``""`` for single-class topics."""

    def __init__(self, discriminator: str, types):
        self.discriminator, self.types = discriminator, list(types)

    def to_c(self) -> CEventJsonTemplate:
        t = CEventJsonTemplate()
        t.n_types = len(self.types)
        t.discriminator = self.discriminator.encode()
        for i, (name, ev_type, seq_field, arg_field, arg_kind) in enumerate(self.types):
            e = t.types[i]
            e.name, e.event_type, e.arg_kind = name.encode(), ev_type, arg_kind
            e.seq_field, e.arg_field = seq_field.encode(), arg_field.encode()
        return t

    def decode(self, value: bytes) -> np.ndarray:
        """One record value -> one ``EVENT_DTYPE`` record (``surge_event_json_decode``)."""
        lib = _native.load()
        out = np.zeros(1, dtype=EVENT_DTYPE)
        c = self.to_c()
        if lib.surge_event_json_validate(ctypes.byref(c)) != 0:
            raise IngestError(-1, (lib.surge_event_json_last_error() or b"").decode())
        buf = (ctypes.c_uint8 * max(len(value), 1)).from_buffer_copy(value or b"\0")
        rc = lib.surge_event_json_decode(ctypes.byref(c), buf, len(value), out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise IngestError(rc, (lib.surge_event_json_last_error() or b"").decode())
        return out[0]


class IngestError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"surge_ingest status {status}: {message}")
        self.status = status


class EventsTopicIngest:
    def __init__(self, isolation_level: int = READ_COMMITTED, frames: bool = False, device_lz4: bool = False, threads: int = 1, device_crc: bool = False):
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        self.frames = frames
        rc = self._lib.surge_ingest_create(isolation_level | (FRAMES if frames else 0) | (DEVICE_LZ4 if device_lz4 else 0) | (DEVICE_CRC if device_crc else 0),
                                           ctypes.byref(self._h))
        if rc != 0:
            raise IngestError(rc, (self._lib.surge_ingest_last_error(None) or b"").decode())
        if (frames or device_lz4 or device_crc) and os.environ.get("SURGE_INGEST_PAGEABLE_ARENA") != "1":
            # page-locked arena when a GPU is there: the device decoder copies out of it in place (a pageable arena works too)
            self._lib.surge_ingest_use_pinned_arena(self._h)
        if threads != 1:  # host threads verifying the batches' CRC-32C of one feed (surge_ingest_set_threads)
            self._check(self._lib.surge_ingest_set_threads(self._h, int(threads)))
        self._tail = b""

    def close(self):
        if self._h:
            self._lib.surge_ingest_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise IngestError(rc, (self._lib.surge_ingest_last_error(self._h) or b"").decode())

    def feed(self, data: bytes) -> int:
        """Decode whole batches; a trailing partial batch is kept and completed by the next call."""
        buf = self._tail + bytes(data) if self._tail else (data if isinstance(data, bytes) else bytes(data))
        consumed = ctypes.c_int64(0)
        # a bytes object is handed over in place (the library copies what it keeps): no per-feed copy of the fetch
        arr = ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p) if buf else None
        rc = self._lib.surge_ingest_feed(self._h, arr, len(buf), ctypes.byref(consumed))
        # also on failure: batches decoded before the failing one ARE queued and must not be fed again
        self._tail = buf[consumed.value:]
        self._check(rc)
        return consumed.value

    @property
    def ready(self) -> int:
        return int(self._lib.surge_ingest_ready(self._h))

    def counters(self) -> dict:
        c = (ctypes.c_int64 * 8)()
        self._check(self._lib.surge_ingest_counters(self._h, ctypes.byref(c)))
        names = ["batches", "records_decoded", "records_delivered", "records_aborted", "control_batches",
                 "flush_records_skipped", "bytes_decompressed", "open_transactions"]
        return dict(zip(names, [int(x) for x in c]))

    def drain_sections(self, max_sections: int = 1 << 30) -> Tuple[np.ndarray, int]:
        """FRAMES mode: the records sections of the deliverable batches (``SECTION_DTYPE`` records whose spans point into
        the arena) and the arena's address — what ``DeviceDecoder.push`` takes."""
        out = np.zeros(min(max_sections, max(self.ready, 1)), dtype=SECTION_DTYPE)  # ready counts records: an upper bound
        got = ctypes.c_int64(0)
        self._check(self._lib.surge_ingest_drain_sections(self._h, out.shape[0], out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(got)))
        return out[: got.value], int(self._lib.surge_ingest_arena(self._h) or 0)

    def drain_fixed16(self, max_records: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        n = self.ready if max_records is None else min(self.ready, max_records)
        agg = np.zeros(n, dtype=np.int64)
        ev = np.zeros(n, dtype=EVENT_DTYPE)
        off = np.zeros(n, dtype=np.int64)
        got = ctypes.c_int64(0)
        self._check(self._lib.surge_ingest_drain_fixed16(
            self._h, n, agg.ctypes.data_as(ctypes.c_void_p), ev.ctypes.data_as(ctypes.c_void_p),
            off.ctypes.data_as(ctypes.c_void_p), ctypes.byref(got)))
        return agg[: got.value], ev[: got.value], off[: got.value]

    def drain_json(self, template: EventJsonTemplate, max_records: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """``(agg_idx, events, offsets)`` for topics whose record values are JSON events, decoded in the library through the
        model's template (``surge_ingest_drain_json``): no per-record Python."""
        n = self.ready if max_records is None else min(self.ready, max_records)
        agg = np.zeros(n, dtype=np.int64)
        ev = np.zeros(n, dtype=EVENT_DTYPE)
        off = np.zeros(n, dtype=np.int64)
        got = ctypes.c_int64(0)
        c = template.to_c()
        self._check(self._lib.surge_ingest_drain_json(
            self._h, n, ctypes.byref(c), agg.ctypes.data_as(ctypes.c_void_p), ev.ctypes.data_as(ctypes.c_void_p),
            off.ctypes.data_as(ctypes.c_void_p), ctypes.byref(got)))
        return agg[: got.value], ev[: got.value], off[: got.value]

    def drain_records(self, max_records: Optional[int] = None) -> List[Tuple[int, int, Optional[bytes], Optional[bytes]]]:
        """``(offset, agg_idx, key, value)`` per record (``None`` for null key / value)."""
        n = self.ready if max_records is None else min(self.ready, max_records)
        recs = np.zeros(n, dtype=RECORD_DTYPE)
        got = ctypes.c_int64(0)
        self._check(self._lib.surge_ingest_drain(self._h, n, recs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(got)))
        base = self._lib.surge_ingest_arena(self._h) or 0  # NULL while the arena is empty (only empty keys / values so far)

        def span(off, ln):
            return None if ln < 0 else (b"" if ln == 0 else ctypes.string_at(base + int(off), int(ln)))

        out = []
        for r in recs[: got.value]:
            key, val = span(r["key_off"], r["key_len"]), span(r["value_off"], r["value_len"])
            out.append((int(r["offset"]), int(r["agg_idx"]), key, val))
        return out

    def key_table(self) -> KeyTable:
        kt = KeyTable()
        n = int(self._lib.surge_ingest_key_count(self._h))
        p, ln = ctypes.c_char_p(), ctypes.c_int64()
        for i in range(n):
            self._check(self._lib.surge_ingest_key(self._h, i, ctypes.byref(p), ctypes.byref(ln)))
            kt.intern(ctypes.string_at(p, ln.value).decode("utf-8"))
        return kt


class FramedFetches:
    """Frames a sequence of fetches ONE FETCH AHEAD of whoever consumes them: a worker thread runs the host stage
    (``feed`` = batch headers, CRC-32C, ``read_committed`` — and the fetch itself, when ``fetches`` is a generator that
    polls) while the caller's thread keeps the device busy with the previous fetch (``DeviceDecoder.push`` + the
    fold).  One ``EventsTopicIngest`` in FRAMES mode does all the framing, so transactions and partial batches carry
    from fetch to fetch; its six rotating arenas (``surge_ingest_drain_sections`` in ``surge_ingest.h``) are what
    makes the overlap safe: the sections of fetch i stay where they are while fetches i + 1 .. i + ``hold`` are framed,
    and the next one is not started before the consumer has asked for fetch i + ``hold`` (= is done with fetch i).
    ``hold`` (1 .. 5) is how many fetches the consumer keeps alive at a time: 1 for push-then-fold, 2 .. 5 when it keeps
    that many ``DeviceDecoder.push_async`` in flight.  ASKING for the next fetch while ``hold`` are held releases the
    oldest one: a consumer with pushes in flight finishes the oldest push first, then asks (``store.restore_from_fetches``).

    Iterating yields ``(sections, arena_address)`` per fetch, in order.  ``overlap=False`` frames inline (same results,
    one thread)."""

    def __init__(self, fetches, isolation_level: int = READ_COMMITTED, device_lz4: bool = True, overlap: bool = True, threads: int = 1, hold: int = 1,
                 device_crc: bool = True):
        import queue
        import threading

        if not 1 <= hold <= 5:
            raise ValueError("hold must be 1 .. 5 (the framer has six arenas)")
        # device_crc needs the sections as they are on the wire: with host-side LZ4 (device_lz4=False) the host has the bytes in hand anyway
        self._g = EventsTopicIngest(isolation_level, frames=True, device_lz4=device_lz4, threads=threads, device_crc=device_crc and device_lz4)
        self._fetches = iter(fetches)
        self._overlap = overlap
        self._hold = hold
        self.framing_seconds: List[float] = []
        self._q: "queue.Queue" = queue.Queue()
        self._slots = threading.Semaphore(hold + 1)  # framed fetches alive at once: the ones being read + the one being framed
        self._stop = False
        self._held = 0
        self._thread = threading.Thread(target=self._run, name="surge-framing", daemon=True) if overlap else None
        if self._thread:
            self._thread.start()

    def _frame(self, data):
        import time

        t0 = time.perf_counter()
        self._g.feed(data)
        item = self._g.drain_sections()
        self.framing_seconds.append(time.perf_counter() - t0)
        return item

    def _run(self):
        try:
            for data in self._fetches:
                self._slots.acquire()
                if self._stop:
                    break
                self._q.put(self._frame(data))
        except BaseException as e:  # handed to the consumer, which re-raises it in its own thread
            self._q.put(e)
        finally:
            self._q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        if not self._overlap:
            return self._frame(next(self._fetches))
        if self._held == self._hold:  # the consumer is done with the oldest fetch it holds: its arena may be framed into again
            self._held -= 1
            self._slots.release()
        item = self._q.get()
        if item is None:
            self._q.put(None)
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(None)
            raise item
        self._held += 1
        return item

    def counters(self) -> dict:
        """The ingest counters; call it when the iteration has ended (the handle belongs to the framing thread)."""
        return self._g.counters()

    def close(self):
        self._stop = True
        if self._thread:
            for _ in range(self._hold + 1):
                self._slots.release()
            self._thread.join()
            self._thread = None
        self._g.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PartitionedFramedFetches:
    """The same one-fetch-ahead framing for a consumer that is assigned several partitions: ``fetches`` yields, per fetch
    response, the next bytes of every partition (a sequence of ``bytes`` / ``None``, always in the same partition order).
    A ``surge_ingest_group`` frames them: one framer per partition — transactions, last stable offsets and cut batches are
    per partition (``SurgeStateStoreConsumer.scala:33-46``: ``isolation.level`` as configured) — side by side on
    ``threads`` host threads (C++ threads inside the library call), all of a fetch's records sections laid out in ONE
    page-locked slab.  Iterating yields ``(sections, slab_address)`` per fetch — partition after partition, offset order
    inside a partition: one ``DeviceDecoder.push_async``, one host-to-device copy.  ``hold`` as in :class:`FramedFetches`
    (the group rotates through six slabs)."""

    def __init__(self, fetches, n_partitions: int, threads: int = 8, hold: int = 3, isolation_level: int = READ_COMMITTED, device_lz4: bool = True,
                 overlap: bool = True, device_crc: bool = True, in_place: bool = False):
        import queue
        import threading

        if not 1 <= hold <= 5:
            raise ValueError("hold must be 1 .. 5 (the group has six slabs)")
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        rc = self._lib.surge_ingest_group_create(n_partitions, isolation_level | (DEVICE_LZ4 if device_lz4 else 0) | (DEVICE_CRC if device_crc and device_lz4 else 0),
                                                 ctypes.byref(self._h))
        if rc != 0:
            raise IngestError(rc, (self._lib.surge_ingest_group_last_error(None) or b"").decode())
        if os.environ.get("SURGE_INGEST_PAGEABLE_ARENA") != "1":
            self._lib.surge_ingest_group_use_pinned_slabs(self._h)  # (fails without a GPU: the slabs stay pageable, which works too)
        self._threads = max(1, min(threads, n_partitions))
        # in_place: the fetch response is RECEIVED into the group's next page-locked slab (surge_ingest_group_receive_copy: one copy out of
        # the bytes objects on the framing threads, standing in for the socket reads a consumer would aim there) and framed
        # where it lies — with device_lz4 and device_crc the host reads the batches' 61-byte headers and nothing else
        self._in_place = in_place and device_lz4
        self.receive_seconds: List[float] = []      # wall time of that stand-in receive copy, per fetch
        self._n = n_partitions
        self._tails = [b""] * n_partitions
        self._fetches = iter(fetches)
        self._overlap, self._hold = overlap, hold
        self.framing_seconds: List[float] = []
        self._ring = [None] * (hold + 2)  # section tables, reused (the consumer keeps `hold` fetches alive while the next is framed)
        self._q: "queue.Queue" = queue.Queue()
        self._slots = threading.Semaphore(hold + 1)
        self._stop = False
        self._held = 0
        self._thread = threading.Thread(target=self._run, name="surge-framing-driver", daemon=True) if overlap else None
        if self._thread:
            self._thread.start()

    def _frame(self, fetch):
        """One library call per fetch (``surge_ingest_group_feed``): no per-partition Python."""
        import time

        t0 = time.perf_counter()
        n = self._n
        fetch = list(fetch)
        if len(fetch) != n:  # (before anything is fed: a short response must not leave some partitions fed and others not)
            raise ValueError(f"a fetch response holds {len(fetch)} entries for {n} partitions (pass None for a partition without bytes)")
        bufs = []
        for tail, data in zip(self._tails, fetch):  # a partition's cut batch from the last fetch goes in front (rare: whole batches are the rule)
            data = data or b""
            bufs.append(tail + bytes(data) if tail else (data if isinstance(data, bytes) else bytes(data)))
        len_arr = (ctypes.c_int64 * n)(*[len(b) for b in bufs])
        if self._in_place:
            tr = time.perf_counter()
            src_arr = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) if b else None for b in bufs])
            data_arr = (ctypes.c_void_p * n)()
            rc = self._lib.surge_ingest_group_receive_copy(self._h, src_arr, len_arr, self._threads, data_arr)
            if rc != 0:
                raise IngestError(rc, (self._lib.surge_ingest_group_last_error(self._h) or b"").decode())
            self.receive_seconds.append(time.perf_counter() - tr)
            t0 = time.perf_counter()  # (framing_seconds: the framing proper)
        else:
            data_arr = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) if b else None for b in bufs])
        # a batch is at least its 61-byte header; a feed also delivers what the partitions still hold from earlier feeds (a
        # transaction whose COMMIT marker only arrives now)
        max_sec = sum(len(b) for b in bufs) // 61 + int(self._lib.surge_ingest_group_queued_sections(self._h)) + 16
        slot = len(self.framing_seconds) % len(self._ring)
        n_sec = ctypes.c_int64()
        slab = ctypes.c_void_p()
        consumed = (ctypes.c_int64 * n)()
        while True:
            if self._ring[slot] is None or self._ring[slot].shape[0] < max_sec:
                self._ring[slot] = np.empty(max_sec + max_sec // 4, dtype=SECTION_DTYPE)
            secs = self._ring[slot]
            rc = self._lib.surge_ingest_group_feed(self._h, data_arr, len_arr, self._threads, consumed, secs.shape[0], secs.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.byref(n_sec), ctypes.byref(slab))
            if rc != 0 and n_sec.value > secs.shape[0]:  # table too small: the feed was undone and says what it needs
                max_sec = int(n_sec.value)
                continue
            break
        if rc != 0:  # all or nothing: the group is what it was before the call (the tails too)
            raise IngestError(rc, (self._lib.surge_ingest_group_last_error(self._h) or b"").decode())
        self._tails = [bufs[p][consumed[p]:] for p in range(n)]
        self.framing_seconds.append(time.perf_counter() - t0)
        return secs[: n_sec.value], int(slab.value or 0)

    _run = FramedFetches._run
    __iter__ = FramedFetches.__iter__
    __next__ = FramedFetches.__next__

    def slab_bytes(self):
        """``(bytes, page_locked)``: what the group's six slabs take (``surge_ingest_group_slab_bytes``)."""
        b, c = ctypes.c_int64(), ctypes.c_int32()
        self._lib.surge_ingest_group_slab_bytes(self._h, ctypes.byref(b), ctypes.byref(c))
        return int(b.value), bool(c.value)

    def cpu_seconds(self):
        """``(receive copy, framing)``: thread CPU seconds the group's host threads have spent so far (``surge_ingest_group_cpu_seconds``)."""
        out = (ctypes.c_double * 2)()
        self._lib.surge_ingest_group_cpu_seconds(self._h, ctypes.byref(out))
        return float(out[0]), float(out[1])

    def counters(self) -> dict:
        """The partitions' ingest counters, summed; call it when the iteration has ended."""
        c = (ctypes.c_int64 * 8)()
        self._lib.surge_ingest_group_counters(self._h, ctypes.byref(c))
        names = ["batches", "records_decoded", "records_delivered", "records_aborted", "control_batches",
                 "flush_records_skipped", "bytes_decompressed", "open_transactions"]
        return dict(zip(names, [int(x) for x in c]))

    def close(self):
        self._stop = True
        if self._thread:
            for _ in range(self._hold + 1):
                self._slots.release()
            self._thread.join()
            self._thread = None
        if self._h:
            self._lib.surge_ingest_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PushPipeline:
    """Keeps ``depth`` pushes in flight with the host work of ENQUEUEING them off the consumer's thread: a worker takes the
    framed fetches from ``source`` (a ``FramedFetches`` / ``PartitionedFramedFetches`` made with ``hold=depth``) and calls
    ``push(item)`` — section tables, staging, a dozen launches: about a millisecond of host time per 10^6-record fetch —
    while the caller's thread finishes the oldest push and folds it.  Iterating yields once per enqueued push (whatever
    ``push`` returned; a falsy return = nothing was pushed for that fetch, nothing is yielded); the caller finishes that
    push and calls ``done()``.  The worker asks ``source`` for fetch i + depth only after ``done()`` for fetch i — asking is
    what lets the framer reuse fetch i's slab.  ``threaded=False``: the same protocol inline, one push at a time."""

    def __init__(self, source, push, depth: int, threaded: bool = True):
        import queue
        import threading

        self._source, self._push, self._depth = iter(source), push, depth
        self._free = threading.Semaphore(depth)
        self._q: "queue.Queue" = queue.Queue()
        self._stop = False
        self.push_seconds: List[float] = []
        self._thread = threading.Thread(target=self._run, name="surge-push", daemon=True) if threaded else None
        if self._thread:
            self._thread.start()

    def _one(self):
        """-> a token to yield, None to skip this fetch, or StopIteration"""
        import time

        item = next(self._source)
        t0 = time.perf_counter()
        token = self._push(item)
        if token:
            self.push_seconds.append(time.perf_counter() - t0)
        return token

    def _run(self):
        try:
            while True:
                self._free.acquire()
                if self._stop:
                    break
                try:
                    token = self._one()
                except StopIteration:
                    break
                if token:
                    self._q.put(token)
                else:
                    self._free.release()
        except BaseException as e:  # handed to the consumer
            self._q.put(e)
        finally:
            self._q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        if self._thread is None:
            while True:
                if not self._free.acquire(blocking=False):
                    raise RuntimeError("PushPipeline(threaded=False): call done() for the previous push before asking for the next")
                token = self._one()  # (StopIteration ends the iteration)
                if token:
                    return token
                self._free.release()
        item = self._q.get()
        if item is None:
            self._q.put(None)
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(None)
            raise item
        return item

    def done(self) -> None:
        self._free.release()

    def close(self) -> None:
        """Stops the worker (it finishes the push it is in).  ``source`` must still be open: the worker may be waiting for
        its next fetch."""
        self._stop = True
        if self._thread:
            for _ in range(self._depth):
                self._free.release()
            self._thread.join()
            self._thread = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DeviceDecoder:
    """``surge_device_decoder``: records sections (host bytes) -> device-resident ``(agg_idx, events, offsets)`` and a
    device key table.  ``template=None``: record values are 16-byte events; otherwise the model's ``EventJsonTemplate``."""

    def __init__(self, template: Optional[EventJsonTemplate] = None, device: int = 0, stream=None):
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        self.device = device
        self._inflight = []
        c = template.to_c() if template is not None else None
        rc = self._lib.surge_device_decoder_create(device, stream, ctypes.byref(c) if c is not None else None, ctypes.byref(self._h))
        if rc != 0:
            raise IngestError(rc, (self._lib.surge_device_decoder_last_error(None) or b"").decode())

    def close(self):
        if self._h:
            self._lib.surge_device_decoder_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise IngestError(rc, (self._lib.surge_device_decoder_last_error(self._h) or b"").decode())

    def push(self, sections: np.ndarray, arena_address: int) -> None:
        sections = np.ascontiguousarray(sections, dtype=SECTION_DTYPE)
        self._check(self._lib.surge_device_decoder_push(self._h, ctypes.c_void_p(arena_address), sections.ctypes.data_as(ctypes.c_void_p),
                                                        sections.shape[0]))

    def push_async(self, parts) -> None:
        """Stage 1 of a push — copy, LZ4, record parsing, value decode — enqueued on a stream of its own; ``parts`` is one
        ``(sections, arena_address)`` pair or a list of them (e.g. one per partition of a fetch response: one push, records
        delivered part after part).  The arenas must stay as they are until the matching ``finish()``; up to five pushes
        may be in flight."""
        if isinstance(parts, tuple):
            parts = [parts]
        secs = [np.ascontiguousarray(s, dtype=SECTION_DTYPE) for s, _ in parts]
        n = len(parts)
        bytes_arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(a) for _, a in parts])
        secs_arr = (ctypes.c_void_p * n)(*[s.ctypes.data_as(ctypes.c_void_p) for s in secs])
        cnt = (ctypes.c_int64 * n)(*[s.shape[0] for s in secs])
        self._check(self._lib.surge_device_decoder_push_parts_async(self._h, n, bytes_arr, secs_arr, cnt))
        self._inflight.append((secs, bytes_arr, secs_arr, cnt))  # (kept until finish(); list.append / pop are atomic: a pushing and a finishing thread share it)

    def finish(self, wait: bool = True) -> None:
        """Stage 2 of the oldest unfinished push: key interning, append to the result (``result()``).  ``wait=False``
        (``surge_device_decoder_push_finish_async``) returns without the closing wait for the device: the results are
        complete in the order of the decoder's stream — hand them over with ``fold_into(engine, wait=False)``.  One thread
        may call ``push_async`` while another calls ``finish`` / ``fold_into`` (``PushPipeline``)."""
        rc = (self._lib.surge_device_decoder_push_finish if wait else self._lib.surge_device_decoder_push_finish_async)(self._h)
        if self._inflight:
            self._inflight.pop(0)
        self._check(rc)

    def fold_into(self, engine, wait: bool = True):
        """Everything decoded since the last clear, folded onto ``engine``'s resident state (grown first for ids seen for the
        first time), and cleared: ``surge_replay_append_decoded``; with ``wait=False`` its ``_async`` form — no host wait,
        the two streams are ordered by events, so the host goes on to enqueue the next push while this fold runs (and with
        the engine on a stream of its own the interning of the next fetch overlaps it on the device too: the decoder then
        rotates stage 1 over two streams instead of three — hardware queues, ``include/surge_ingest.h``).  What the fold
        has to report it reports at the engine's next ``synchronize``.  Returns ``(n_events, n_keys)``."""
        n_ev, n_keys = ctypes.c_int64(), ctypes.c_int64()
        fn = self._lib.surge_replay_append_decoded if wait else self._lib.surge_replay_append_decoded_async
        self._check(fn(engine._h, self._h, ctypes.byref(n_ev), ctypes.byref(n_keys)))
        return int(n_ev.value), int(n_keys.value)

    def stage_into(self, engine):
        """Everything decoded since the last clear, appended to ``engine``'s staging log instead of folded, and cleared
        (``surge_replay_stage_decoded``; no host wait).  A recovery that folds the topic once ends with
        ``engine.pack_staged(n_keys)`` + one ``engine.fold()``.  Returns ``(n_events, n_keys)``."""
        n_ev, n_keys = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.surge_replay_stage_decoded(engine._h, self._h, ctypes.byref(n_ev), ctypes.byref(n_keys)))
        return int(n_ev.value), int(n_keys.value)

    @property
    def n_keys(self) -> int:
        n = ctypes.c_int64()
        self._check(self._lib.surge_device_decoder_result(self._h, None, None, None, None, ctypes.byref(n)))
        return int(n.value)

    def reserve(self, n_keys: int, key_bytes: int) -> None:
        """Capacity hint (``surge_device_decoder_reserve``): room for ``n_keys`` aggregate ids of ``key_bytes`` bytes in all."""
        self._check(self._lib.surge_device_decoder_reserve(self._h, int(n_keys), int(key_bytes)))

    @property
    def pending(self) -> int:
        return int(self._lib.surge_device_decoder_pending(self._h))

    def push_records(self, keys: List[bytes], values: List[bytes], offsets=None) -> None:
        """Records that are already framed (a consumer's key / value byte arrays), in bulk."""
        def table(items):
            off = np.zeros(len(items) + 1, dtype=np.int64)
            if items:
                np.cumsum([len(x) for x in items], out=off[1:])
            data = np.frombuffer(b"".join(items), dtype=np.uint8) if off[-1] else np.zeros(1, np.uint8)
            return data, off

        kd, ko = table(keys)
        vd, vo = table(values)
        of = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        self._check(self._lib.surge_device_decoder_push_records(
            self._h, kd.ctypes.data_as(ctypes.c_void_p), ko.ctypes.data_as(ctypes.c_void_p), vd.ctypes.data_as(ctypes.c_void_p),
            vo.ctypes.data_as(ctypes.c_void_p), of.ctypes.data_as(ctypes.c_void_p) if of is not None else None, len(keys)))

    def push_from(self, ingest: EventsTopicIngest) -> int:
        """Everything ``ingest`` (created with ``frames=True``) can deliver now; returns the number of batches."""
        sections, arena = ingest.drain_sections()
        if sections.shape[0]:
            self.push(sections, arena)
        return int(sections.shape[0])

    def result(self):
        """``(agg_idx, events, offsets)`` as CUDA tensors viewing the decoder's arrays (valid until the next push / clear)
        and the number of keys so far."""
        import torch

        n, nk = ctypes.c_int64(), ctypes.c_int64()
        pa, pe, po = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.surge_device_decoder_result(self._h, ctypes.byref(n), ctypes.byref(pa), ctypes.byref(pe), ctypes.byref(po), ctypes.byref(nk)))
        dev = torch.device("cuda", self.device)
        if n.value == 0:
            z = torch.zeros(0, dtype=torch.int64, device=dev)
            return z, torch.zeros((0, 2), dtype=torch.int64, device=dev), z.clone(), nk.value

        def view(ptr, count, cols):
            iface = {"shape": (count, cols) if cols > 1 else (count,), "typestr": "<i8", "data": (ptr.value, False), "version": 2}
            holder = type("_Span", (), {"__cuda_array_interface__": iface})()
            return torch.as_tensor(holder, device=dev)

        return view(pa, n.value, 1), view(pe, n.value, 2), view(po, n.value, 1), nk.value

    def clear(self) -> None:
        self._check(self._lib.surge_device_decoder_clear(self._h))

    def keys(self) -> List[str]:
        n, nb = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.surge_device_decoder_keys(self._h, None, 0, None, ctypes.byref(n), ctypes.byref(nb)))
        data = np.zeros(max(nb.value, 1), np.uint8)
        off = np.zeros(n.value + 1, np.int64)
        self._check(self._lib.surge_device_decoder_keys(self._h, data.ctypes.data_as(ctypes.c_void_p), data.shape[0], off.ctypes.data_as(ctypes.c_void_p),
                                                        ctypes.byref(n), ctypes.byref(nb)))
        raw = data.tobytes()
        return [raw[off[i]:off[i + 1]].decode("utf-8") for i in range(n.value)]

    def counters(self) -> dict:
        c = (ctypes.c_int64 * 4)()
        self._lib.surge_device_decoder_counters(self._h, ctypes.byref(c))
        return dict(zip(["records_seen", "records_delivered", "flush_records_skipped", "doubles_parsed_on_host"], [int(x) for x in c]))

    def stats(self) -> dict:
        """The counters plus the key table's: re-seeds after a detected 64-bit hash collision, slots, pushes, hash function."""
        c = (ctypes.c_int64 * 8)()
        self._lib.surge_device_decoder_stats(self._h, ctypes.byref(c))
        return dict(zip(["records_seen", "records_delivered", "flush_records_skipped", "doubles_parsed_on_host", "hash_reseeds", "table_slots", "pushes",
                         "hash_function"], [int(x) for x in c]))
