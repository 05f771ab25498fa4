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

import ctypes
import time
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
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


class RecordBatchWriter:
    """Kafka record batches (message format v2) of a state topic, one log per partition, through the C ABI of
    ``include/surge_snapshot.h`` (host C++; the same library decodes them again in ``include/surge_ingest.h``)."""

    CODECS = {"none": 0, "lz4": 3}  # Kafka attribute bits 0-2; lz4 is the reference producer's setting (reference.conf:112)

    def __init__(self, n_partitions: int, max_records_per_batch: int = 0, max_batch_bytes: int = 0, compression: str = "none"):
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        rc = self._lib.surge_snapshot_writer_create(n_partitions, max_records_per_batch, max_batch_bytes, ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"surge_snapshot_writer_create: {rc}: {(self._lib.surge_snapshot_writer_last_error(None) or b'').decode()}")
        self.n_partitions = n_partitions
        if compression != "none":
            self.set_compression(compression)

    def set_compression(self, compression: str) -> None:
        """Codec of the batches closed from now on (call right after create, flush or reset)."""
        self._check(self._lib.surge_snapshot_writer_set_compression(self._h, self.CODECS[compression]))

    def close(self):
        if self._h:
            self._lib.surge_snapshot_writer_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"surge_snapshot_writer: {rc}: {(self._lib.surge_snapshot_writer_last_error(self._h) or b'').decode()}")

    def append(self, kind, partition, keys_utf8, key_off, values, val_off, timestamp_ms: Optional[int] = None) -> None:
        """One record per index with ``kind != SKIP`` (``kind=None``: all values); numpy arrays, no per-record Python."""
        p = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)  # noqa: E731
        kind, partition = p(kind, np.uint8), p(partition, np.int32)
        keys_utf8, key_off, values, val_off = p(keys_utf8, np.uint8), p(key_off, np.int64), p(values, np.uint8), p(val_off, np.int64)
        n = partition.shape[0]
        ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        self._check(self._lib.surge_snapshot_writer_append(
            self._h, n, ptr(kind), ptr(partition), ptr(keys_utf8), ptr(key_off), ptr(values), ptr(val_off),
            int(time.time() * 1000) if timestamp_ms is None else int(timestamp_ms)))

    def append_indexed(self, agg_idx, kind, partition, keys_utf8, key_off, values, val_off, timestamp_ms: Optional[int] = None) -> None:
        """Compact form: record r is aggregate ``agg_idx[r]``; ``kind`` / ``val_off`` per record, ``partition`` / ``key_off`` per aggregate."""
        p = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)  # noqa: E731
        agg_idx, kind, partition = p(agg_idx, np.int64), p(kind, np.uint8), p(partition, np.int32)
        keys_utf8, key_off, values, val_off = p(keys_utf8, np.uint8), p(key_off, np.int64), p(values, np.uint8), p(val_off, np.int64)
        ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        self._check(self._lib.surge_snapshot_writer_append_indexed(
            self._h, agg_idx.shape[0], ptr(agg_idx), partition.shape[0], ptr(kind), ptr(partition), ptr(keys_utf8), ptr(key_off), ptr(values), ptr(val_off),
            int(time.time() * 1000) if timestamp_ms is None else int(timestamp_ms)))

    def partition_bytes(self, partition: int) -> Tuple[bytes, int, int]:
        """``(record batches, records, next offset)`` of one partition's log (flushes the open batches)."""
        self._check(self._lib.surge_snapshot_writer_flush(self._h))
        data, ln, nrec, nxt = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.surge_snapshot_writer_partition(self._h, partition, ctypes.byref(data), ctypes.byref(ln), ctypes.byref(nrec), ctypes.byref(nxt)))
        return (ctypes.string_at(data, ln.value) if ln.value else b""), nrec.value, nxt.value

    def reset(self) -> None:
        self._check(self._lib.surge_snapshot_writer_reset(self._h))


class DeviceFramer:
    """Record batches of a bulk publish framed ON THE GPU (``surge_device_framer_*`` in ``include/surge_snapshot.h``): the
    inputs are the device arrays a publish already has (the delta's kinds, the encoder's text + offsets, the key table,
    the partitions); the output is byte-identical to ``RecordBatchWriter.append`` + flush on the same input
    (uncompressed batches).  Needs a GPU."""

    def __init__(self, n_partitions: int, device: int = 0, max_records_per_batch: int = 0, max_batch_bytes: int = 0):
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        self.n_partitions = n_partitions
        rc = self._lib.surge_device_framer_create(device, None, n_partitions, max_records_per_batch, max_batch_bytes, ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"surge_device_framer_create: {rc}: {(self._lib.surge_device_framer_last_error(None) or b'').decode()}")
        self.records = self.batches = 0

    def close(self):
        if self._h:
            self._lib.surge_device_framer_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def frame(self, d_kind, d_partition, d_keys, d_key_off, d_values, d_val_off, timestamp_ms: Optional[int] = None) -> Dict[int, memoryview]:
        """``{partition: record batches}`` for the aggregates with ``kind != SKIP``; CUDA tensors in (all per aggregate),
        zero-copy views of the framer's page-locked buffer out — valid until the next call (``bytes(v)`` to keep one)."""
        n = int(d_kind.numel())
        if int(d_partition.numel()) < n or int(d_key_off.numel()) < n + 1 or (d_val_off is not None and int(d_val_off.numel()) < n + 1):
            raise ValueError("partition / key_off / val_off are shorter than kind")
        ptr = lambda t: None if t is None or int(t.numel()) == 0 else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        data, offs = ctypes.c_void_p(), ctypes.c_void_p()
        nrec, nbat = ctypes.c_int64(), ctypes.c_int64()
        rc = self._lib.surge_device_framer_frame(self._h, n, ptr(d_kind), ptr(d_partition), ptr(d_keys), ptr(d_key_off), ptr(d_values), ptr(d_val_off),
                                                 int(time.time() * 1000) if timestamp_ms is None else int(timestamp_ms),
                                                 ctypes.byref(data), ctypes.byref(offs), ctypes.byref(nrec), ctypes.byref(nbat))
        if rc != 0:
            raise RuntimeError(f"surge_device_framer_frame: {rc}: {(self._lib.surge_device_framer_last_error(self._h) or b'').decode()}")
        self.records, self.batches = nrec.value, nbat.value
        off = np.ctypeslib.as_array(ctypes.cast(offs, ctypes.POINTER(ctypes.c_int64)), shape=(self.n_partitions + 1,)).copy()
        total = int(off[-1])
        if total == 0:
            return {}
        whole = memoryview((ctypes.c_uint8 * total).from_address(data.value)).cast("B")
        return {p: whole[int(off[p]):int(off[p + 1])] for p in range(self.n_partitions) if off[p + 1] > off[p]}

    def next_offsets(self) -> np.ndarray:
        out = np.zeros(self.n_partitions, dtype=np.int64)
        self._lib.surge_device_framer_next_offsets(self._h, out.ctypes.data_as(ctypes.c_void_p))
        return out


class BulkSnapshotPublisher:
    """State-topic record batches straight from the GPU-resident states, no per-aggregate host work (N2 x N3):

        delta kernel (what changed since the last publish: PersistentActor.scala:212,257)
          -> GPU encoder restricted to the changed Some aggregates (writeState text; N3)
          -> K4 partitions of the aggregate ids (once per key table)
          -> uncompressed batches: DeviceFramer (records + headers written on the GPU, one D2H, CRC-32C on the host)
             lz4 batches: one D2H of {kinds, text, offsets} -> RecordBatch v2 encoder (C++; N2)
          -> bytes per partition (device framing: views of a page-locked buffer, valid until the next publish).

    ``keys`` are the aggregate ids in dense-index order; ``template`` declares the model's serialized state."""

    def __init__(self, engine, keys: Optional[Sequence[str]], n_partitions: int, template=None, device=None, tables=None,
                 compression: str = "none", device_framing: bool = True):
        """``keys``: aggregate ids in dense-index order; or ``tables = (keys_utf8, key_off, keys_utf16, off16)`` as
        tensors / arrays built without Python strings (large synthetic populations).  ``compression``: "none" or "lz4"
        (the reference producer's ``compression.type``)."""
        import torch

        from .encode import JsonTemplate, key_table_utf8
        from .kafka import utf16_table

        self.engine, self.n_partitions = engine, n_partitions
        self.template = template or JsonTemplate.counter()
        self.device = torch.device(device or f"cuda:{engine.device}")
        self.writer = RecordBatchWriter(n_partitions, compression=compression)
        if tables is None:
            data, off = key_table_utf8(keys)
            u16, o16 = utf16_table(keys)
            u16 = u16.view(np.int16)
        else:
            data, off, u16, o16 = tables
        as_t = lambda a: a.to(self.device) if hasattr(a, "to") else torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.d_keys, self.d_key_off = as_t(data), as_t(off)
        self.h_keys, self.h_key_off = self.d_keys.cpu().numpy(), self.d_key_off.cpu().numpy()
        n_keys = int(self.d_key_off.numel()) - 1
        d_part = torch.zeros(n_keys, dtype=torch.int32, device=self.device)
        if n_keys:
            torch.cuda.current_stream(self.device).synchronize()
            engine.partition_hash_device(as_t(u16), as_t(o16), n_partitions, d_part, up_to_colon=True)
            engine.synchronize()
        self.partitions = d_part.cpu().numpy()
        self.d_part = d_part
        # uncompressed batches are framed on the device (DeviceFramer: byte-identical to the host writer); lz4 batches
        # go through the host writer, whose compressor it is
        self.framer = DeviceFramer(n_partitions, device=self.device.index or 0) if compression == "none" and device_framing else None
        self.timings: Dict[str, float] = {}

    def close(self):
        self.writer.close()
        if self.framer is not None:
            self.framer.close()

    def publish(self, commit: bool = True, timestamp_ms: Optional[int] = None) -> Dict[int, bytes]:
        """Record batches (per partition) for everything that changed since the last committed publish.

        The baseline moves only AFTER the batches exist (delta with commit = 0, encode, copy, frame, then
        ``surge_replay_snapshot_commit``): a failure anywhere on the way leaves the changed aggregates unpublished, to be
        emitted again by the next call.  ``commit=False`` defers that last step to ``commit_published()`` — call it when
        the producer acknowledged the records (the reference's notion of "published")."""
        import torch

        from .encode import encode_states

        eng = self.engine
        n = eng.n_agg
        if n > len(self.partitions):
            raise ValueError("the resident state has aggregates the publisher has no key for: rebuild it with the current key table")
        lib = _native.load()
        t0 = time.perf_counter()
        d_kind = torch.zeros(n, dtype=torch.uint8, device=self.device)
        nv, nt = ctypes.c_int64(), ctypes.c_int64()
        eng._check(lib.surge_replay_snapshot_delta(eng._h, ctypes.c_void_p(d_kind.data_ptr()), ctypes.byref(nv), ctypes.byref(nt), 0))
        self._pending_kind = None
        eng._check(lib.surge_replay_set_encode_filter(eng._h, ctypes.c_void_p(d_kind.data_ptr())))
        try:
            d_out, d_off = encode_states(eng, self.template, self.d_keys, self.d_key_off, capacity_hint=max(64, 96 * nv.value + int(self.d_keys.numel())))
        finally:
            eng._check(lib.surge_replay_set_encode_filter(eng._h, None))
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        if self.framer is not None:
            # the records and batch headers are written on the device; the host only adds the batches' CRCs
            out = self.framer.frame(d_kind, self.d_part, self.d_keys, self.d_key_off, d_out, d_off, timestamp_ms)
            self._pending_kind = d_kind
            if commit:
                self.commit_published()
            t3 = time.perf_counter()
            self.timings = {"gpu_delta_and_encode_ms": (t1 - t0) * 1e3, "device_framing_copy_crc_ms": (t3 - t1) * 1e3, "values": nv.value,
                            "tombstones": nt.value, "text_bytes": int(d_off[n].item()), "batches": self.framer.batches}
            return out
        # compact on the device: only what changed crosses PCIe and is walked by the writer (the encoder wrote text for
        # VALUE aggregates only, so the text is already contiguous and the selected offsets are its record boundaries)
        sel = torch.nonzero(d_kind).squeeze(1)
        kind = d_kind[sel].cpu().numpy()
        off = torch.cat((d_off[sel], d_off[n:n + 1])).cpu().numpy()
        text, sel_h = d_out.cpu().numpy(), sel.cpu().numpy()
        t2 = time.perf_counter()
        self.writer.reset()
        self.writer.append_indexed(sel_h, kind, self.partitions[:n], self.h_keys, self.h_key_off[: n + 1], text, off, timestamp_ms)
        out = {}
        for p in range(self.n_partitions):
            data, nrec, _ = self.writer.partition_bytes(p)
            if nrec:
                out[p] = data
        self._pending_kind = d_kind
        if commit:
            self.commit_published()
        t3 = time.perf_counter()
        self.timings = {"gpu_delta_and_encode_ms": (t1 - t0) * 1e3, "d2h_ms": (t2 - t1) * 1e3, "record_batches_ms": (t3 - t2) * 1e3,
                        "values": nv.value, "tombstones": nt.value, "text_bytes": int(text.nbytes)}
        return out

    def publish_async(self, timestamp_ms: Optional[int] = None) -> "PendingPublish":
        """The same publish with the host framing in the BACKGROUND, so that the store keeps folding micro-batches meanwhile:
        the GPU part (delta, encode, compaction, device -> host copy) runs now and commits the baseline — it is exactly what
        was encoded —; the record batches are framed on a worker thread.  ``result()`` of the returned handle gives them;
        should the framing fail, the baseline of the reported aggregates is invalidated (``surge_replay_snapshot_invalidate``)
        and the next publish emits them again.  One publish may be pending at a time (a second call waits for the first)."""
        import threading

        import torch

        from .encode import encode_states

        prev = getattr(self, "_pending_publish", None)
        if prev is not None:
            prev.wait()
        if self.framer is not None:
            # device framing leaves the host a few milliseconds of CRCs: nothing worth a thread, and one log of offsets
            out = self.publish(commit=True, timestamp_ms=timestamp_ms)
            pending = PendingPublish(self, None, dict(self.timings))
            pending._out = {p: bytes(v) for p, v in out.items()}  # outlives the framer's buffer, like the writer's bytes
            self._pending_publish = pending
            return pending
        eng = self.engine
        n = eng.n_agg
        if n > len(self.partitions):
            raise ValueError("the resident state has aggregates the publisher has no key for: rebuild it with the current key table")
        lib = _native.load()
        t0 = time.perf_counter()
        d_kind = torch.zeros(n, dtype=torch.uint8, device=self.device)
        nv, nt = ctypes.c_int64(), ctypes.c_int64()
        eng._check(lib.surge_replay_snapshot_delta(eng._h, ctypes.c_void_p(d_kind.data_ptr()), ctypes.byref(nv), ctypes.byref(nt), 1))
        eng._check(lib.surge_replay_set_encode_filter(eng._h, ctypes.c_void_p(d_kind.data_ptr())))
        try:
            d_out, d_off = encode_states(eng, self.template, self.d_keys, self.d_key_off, capacity_hint=max(64, 96 * nv.value + int(self.d_keys.numel())))
        except Exception:
            eng._check(lib.surge_replay_set_encode_filter(eng._h, None))
            eng._check(lib.surge_replay_snapshot_invalidate(eng._h, ctypes.c_void_p(d_kind.data_ptr())))
            raise
        eng._check(lib.surge_replay_set_encode_filter(eng._h, None))
        sel = torch.nonzero(d_kind).squeeze(1)
        kind = d_kind[sel].cpu().numpy()
        off = torch.cat((d_off[sel], d_off[n:n + 1])).cpu().numpy()
        text, sel_h = d_out.cpu().numpy(), sel.cpu().numpy()
        pending = PendingPublish(self, d_kind, {"gpu_and_copy_ms": (time.perf_counter() - t0) * 1e3, "values": nv.value, "tombstones": nt.value,
                                                "text_bytes": int(text.nbytes)})

        def frame():
            try:
                t1 = time.perf_counter()
                self.writer.reset()
                self.writer.append_indexed(sel_h, kind, self.partitions[:n], self.h_keys, self.h_key_off[: n + 1], text, off, timestamp_ms)
                out = {}
                for p in range(self.n_partitions):
                    data, nrec, _ = self.writer.partition_bytes(p)
                    if nrec:
                        out[p] = data
                pending.timings["record_batches_ms"] = (time.perf_counter() - t1) * 1e3
                pending._out = out
            except BaseException as exc:  # noqa: BLE001  (handed to result())
                pending._exc = exc

        pending._thread = threading.Thread(target=frame, name="surge-snapshot-framing", daemon=True)
        pending._thread.start()
        self._pending_publish = pending
        return pending

    def commit_published(self) -> None:
        """Make what the last ``publish(commit=False)`` reported the new baseline (``surge_replay_snapshot_commit``).

        The commit copies the aggregates' states as they are *now*, so nothing may fold, append or grow on the engine
        between that ``publish`` and this call: the library refuses with ``SURGE_E_STATE`` (``ReplayError``) if it did
        — publish again.  A publisher that wants the store to keep folding while the producer acknowledges uses
        ``publish_async`` (baseline = exactly what was encoded, invalidated if the records never make it out)."""
        if getattr(self, "_pending_kind", None) is None:
            return
        eng = self.engine
        eng._check(_native.load().surge_replay_snapshot_commit(eng._h, ctypes.c_void_p(self._pending_kind.data_ptr())))
        self._pending_kind = None


class PendingPublish:
    """A ``publish_async`` whose record batches are being framed on a worker thread."""

    def __init__(self, publisher: "BulkSnapshotPublisher", d_kind, timings: dict):
        self._publisher, self._d_kind, self.timings = publisher, d_kind, timings
        self._thread = None
        self._out: Optional[Dict[int, bytes]] = None
        self._exc: Optional[BaseException] = None
        self._settled = False

    def wait(self) -> None:
        if self._thread is not None:
            self._thread.join()
        if self._exc is not None and not self._settled:
            eng = self._publisher.engine  # the records never existed: their baseline must not stand
            eng._check(_native.load().surge_replay_snapshot_invalidate(eng._h, ctypes.c_void_p(self._d_kind.data_ptr())))
        self._settled = True
        self._d_kind = None

    def result(self) -> Dict[int, bytes]:
        self.wait()
        if self._exc is not None:
            raise self._exc
        return self._out


def compact(records: Iterable[StateRecord]) -> Dict[str, Optional[bytes]]:
    """Kafka log compaction / KTable materialisation: last record per key wins; tombstones delete."""
    table: Dict[str, Optional[bytes]] = {}
    for r in records:
        if r.value is None:
            table.pop(r.key, None)
        else:
            table[r.key] = r.value
    return table
