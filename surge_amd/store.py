"""GPU-backed aggregate state store behind the reference's read seams.

* S2 ``getAggregateBytes(aggregateId): Future[Option[Array[Byte]]]`` —
  ``modules/common/src/main/scala/surge/kafka/streams/AggregateStateStoreKafkaStreams.scala:83-85``,
  called once per actor start by ``KTableInitializationSupport.fetchState``
  (``modules/command-engine/core/src/main/scala/surge/internal/persistence/KTableInitializationSupport.scala:63-74``).
* S1 ``SurgeKafkaStreamsPersistencePlugin.createSupplier(storeName)`` —
  ``.../streams/SurgeKafkaStreamsPersistencePlugin.scala:12-15``: the key/value store the KTable
  topology writes state-topic records into (last value per key wins, ``null`` deletes —
  ``SurgeStateStoreConsumer.scala:69``).

The reference rebuilds this store by replaying the compacted *state* topic into RocksDB; here the
same bytes are produced by folding the *events* topic on the GPU (equal by the reference's own
invariant: events and snapshot are published in one transaction).  The serialized
format stays the plugin's: the engine hands a fixed-width state to ``writeState``.
"""
from __future__ import annotations

import bisect
import contextlib
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .command import ReplayableCommandModel, SurgeCommandBusinessLogic
from .log import EventLog, KeyTable, pack_events
from .replay import ReplayEngine
from .schema import ALGO_AUTO, STATE_DTYPE, STATE_POISONED, STATE_PRESENT


class AggregateInitializationException(RuntimeError):
    """Mirror of the error an actor gets when its state cannot be initialised
    (``PersistentActor.scala:328-333``); raised for aggregates whose replay hit a throwing event."""


class GpuReplayStateStore:
    """Recover aggregate state by GPU replay and serve it through ``get_aggregate_bytes``."""

    def __init__(self, business_logic: SurgeCommandBusinessLogic, device: int = 0):
        model = business_logic.command_model()
        if not isinstance(model, ReplayableCommandModel):
            raise TypeError("GPU replay needs a ReplayableCommandModel (event algebra + fixed-width codecs)")
        self.business_logic = business_logic
        self.model = model
        self.engine = ReplayEngine(model.event_algebra(), device)
        self.keys = KeyTable()
        self.store_name = f"{business_logic.aggregate_name}AggregateStateStore"  # SurgeStateStoreConsumer.scala:110
        self._restored = False

    def close(self):
        self.engine.close()

    # -- recovery ---------------------------------------------------------------------------
    def restore(self, events_in_offset_order: Sequence, prior: Optional[Dict[str, object]] = None,
                capacity: int = 0, algo: int = ALGO_AUTO) -> None:
        """Fold the whole events topic.  ``prior`` (id -> aggregate) is an earlier snapshot to fold onto."""
        if prior:
            for k in prior:
                self.keys.intern(k)
        log = pack_events(self.model, events_in_offset_order, self.keys, capacity)
        init = None
        if prior:
            init = np.zeros(log.n_aggregates, dtype=STATE_DTYPE)
            for k, agg in prior.items():
                init[self.keys.index[k]] = self.model.state_to_fixed(agg)[0]
        self.restore_log(log, init, algo)

    def restore_from_topic(self, record_batches: bytes, capacity: int = 0, algo: int = ALGO_AUTO) -> dict:
        """Recover from the raw bytes of one events-topic partition (Kafka record batches v2, lz4 or none,
        ``read_committed``): ingest (``include/surge_ingest.h``) -> CSR pack -> GPU fold.

        Record values that are exactly 16 bytes are taken as fixed-width events; the plugin's JSON event text is decoded
        through the model's ``event_json_template``.  Both go the device way: the host only frames the batches (headers,
        CRC, transactions, LZ4), the GPU parses the records, interns the aggregate ids, decodes the values and groups
        them (``surge_device_decoder`` + the K3 group-by) — no per-record work in any host language.  A model without a
        template falls back to the host decoder and the plugin's ``SurgeEventReadFormatting.read_event`` per record.
        Returns the ingest counters."""
        from .core import SerializedMessage
        from .ingest import EventsTopicIngest, IngestError
        from .schema import EVENT_DTYPE

        template = self.model.event_json_template()
        try:
            return self.restore_from_fetches([record_batches], capacity=capacity, overlap=False)
        except _NotDeviceDecodable:
            pass  # JSON values and a model without a template: the host decoder + the plugin's reader below
        with EventsTopicIngest() as g:
            g.feed(record_batches)
            recs = None
            try:
                # GPU-ready topic (every value IS the 16-byte fixed event): one vectorised drain, no per-record Python
                agg_idx, events, _ = g.drain_fixed16()
            except IngestError:
                if template is not None:
                    # the plugin's JSON event text, decoded in the library through the model's template
                    agg_idx, events, _ = g.drain_json(template)
                else:
                    recs = g.drain_records()  # no template: every value through the plugin's own event reader
            keys = g.key_table()
            counters = g.counters()
        if recs is not None:
            reader = self.business_logic.event_write_formatting()
            events = np.zeros(len(recs), dtype=EVENT_DTYPE)
            agg_idx = np.zeros(len(recs), dtype=np.int64)
            for i, (_, idx, key, value) in enumerate(recs):
                agg_idx[i] = idx
                if value is not None and len(value) == 16:
                    events[i] = np.frombuffer(value, dtype=EVENT_DTYPE)[0]
                else:
                    evt = reader.read_event(SerializedMessage(key.decode("utf-8"), value or b""))
                    events[i] = self.model.encode_events([evt])[0]
        # The records are in topic (offset) order, tagged with their aggregate's dense index: exactly a micro-batch.  The
        # group-by runs on the device (stable radix sort + head scan, the K3 path) onto an all-None resident state; no
        # host-side sort of the whole topic.
        n_agg = max(len(keys), capacity)
        self.keys = keys
        self.engine.load_csr(np.zeros(n_agg + 1, dtype=np.int64), np.zeros(0, dtype=EVENT_DTYPE))
        self.engine.fold()  # every aggregate None; `algo` only names kernels for bound logs (restore / restore_log)
        if events.shape[0]:
            self.engine.append_events(agg_idx, events)
        self.engine.snapshot()  # publishes the host mirror that serves point reads
        self._restored = True
        return counters

    def restore_from_fetches(self, fetches, capacity: int = 0, overlap: bool = True, n_partitions: int = 0, framing_threads: int = 8,
                             consumer_threads: int = 1, bound_log: bool = False, algo: int = ALGO_AUTO, device_crc: bool = True, in_place: bool = True) -> dict:
        """Recover from the events topic as a consumer receives it: ``fetches`` yields the record-batch bytes of one
        partition, fetch by fetch, in offset order (a list, or a generator that polls) — or, with ``n_partitions``, per
        fetch response the next bytes of each of the consumer's partitions (``PartitionedFramedFetches``: one framer per
        partition on ``framing_threads`` host threads, one device push per fetch).  The host frames the next fetches
        (headers, CRC-32C, transactions) while the GPU works on up to four earlier ones: their copy, LZ4 blocks, record
        parsing and value decode (``DeviceDecoder.push_async``) run ahead of the one whose keys are being interned and
        whose events are grouped and folded onto the resident state (the K3 path), so neither side waits for the other;
        the resident state and the device key table grow as new aggregates appear.  What
        ``SurgeStateStoreConsumer.scala:33-46,57-76`` does record by record through Kafka Streams' restore.
        One consumer thread enqueues a push, waits for the interning of the oldest one and hands its events to the fold by an
        event (``surge_replay_append_decoded_async``): no host wait behind the fold.  ``consumer_threads=2`` moves the host
        work of enqueueing a push (section tables, staging, launches: 0.3 ms per 10^6-record fetch) to a worker thread
        (``PushPipeline``) and drops the wait behind the interning too; measured on the 10^7-aggregate topic it is not
        faster, so one thread is the default.
        ``bound_log=True``: a recovery that folds ONCE — every fetch's decoded events are staged on the device instead of
        folded (``surge_replay_stage_decoded``), the topic's end packs them into one CSR log (``surge_replay_pack_staged``:
        stable device sort by aggregate, topic order kept inside an aggregate) and ONE fold with ``algo`` (AUTO: the
        lane-per-row kernels for a log large enough) produces the states; the packed log stays bound, so ``engine.fold`` /
        ``engine.prepare`` can replay it.  What the reference's restore gets from RocksDB's key order
        (``SurgeStateStoreConsumer.scala:57-76``).
        Needs values the device decoder reads (16-byte fixed events, or JSON with the model's
        ``event_json_template``); returns the ingest + decoder counters."""
        from .ingest import DeviceDecoder, FramedFetches, PartitionedFramedFetches, PushPipeline
        from .log import KeyTable
        from .schema import EVENT_DTYPE

        template = self.model.event_json_template()
        d = None
        n_agg = -1
        # pushes in flight (measured on the 10 M-aggregate topic: 2 / 3 / 4 / 5 in flight = 4.1 / 4.3 / 4.65 / 4.7e8 events/s on
        # one box): stage 1 of the next fetches runs while this one is interned and folded
        depth = 4 if overlap else 1

        def push(item):
            """(the pipeline's worker thread) stage 1 of one fetch; the decoder is made for the first fetch with a record in it"""
            nonlocal d
            parts = item if isinstance(item, list) else [item]
            parts = [(sec, arena) for sec, arena in parts if sec.shape[0]]
            if not parts and d is None:
                return False  # (nothing pushed yet: nothing in flight that the framer could overtake)
            if d is None:
                kind = None
                for sec, arena in parts:
                    kind = kind or _sniff_value_kind(sec, arena)  # "fixed16", "json" or None (no deliverable record)
                if kind == "json" and template is None:
                    raise _NotDeviceDecodable("JSON event values need the model's event_json_template")
                d = DeviceDecoder(template if kind == "json" else None, device=self.engine.device)
            d.push_async(parts)
            return True

        two_threads = overlap and consumer_threads >= 2

        def finish_one():
            nonlocal n_agg
            if bound_log:
                d.finish(wait=not two_threads)
                _, n_keys = d.stage_into(self.engine)  # no fold: the topic's end packs what was staged
                n_agg = max(n_agg, n_keys)
                return
            if n_agg < 0:
                n_agg = capacity
                self.engine.load_csr(np.zeros(n_agg + 1, dtype=np.int64), np.zeros(0, dtype=EVENT_DTYPE))
                self.engine.fold()  # every aggregate None
            # No host wait behind the fold: the decoder's stream and the engine's are ordered by events, so the next push is
            # enqueued — and, with two threads (no wait behind the interning either), the next fetch interned — beside this
            # one's group-by and fold (one thread: 7.7 -> 8.7e8 events/s, profiles/r06_e2e_consumer_waits_queues2.txt).  What a
            # fold has to report (a bad index, a poisoned state) it reports at the synchronisation that ends the restore.
            d.finish(wait=not two_threads)
            _, n_keys = d.fold_into(self.engine, wait=False)  # grows the resident state for new ids, group-by + fold (K3), clears
            if n_keys > n_agg:
                self.engine.n_agg = n_agg = n_keys  # (grown inside the call)

        try:
            # device_crc: the batches' CRC-32C is finished on the GPU where their bytes go anyway (SURGE_INGEST_DEVICE_CRC): the framing
            # threads touch a batch's header, not its bytes
            # in_place: a fetch response is received into the framer's page-locked slab and framed where it lies (no copy of the sections)
            framer = (PartitionedFramedFetches(fetches, n_partitions, threads=framing_threads, hold=depth, overlap=overlap, device_crc=device_crc, in_place=in_place)
                      if n_partitions
                      else FramedFetches(fetches, overlap=overlap, hold=depth, device_crc=device_crc))
            with framer as framed, (self.engine.on_own_stream() if two_threads else contextlib.nullcontext()):
                try:
                    if two_threads:
                        # the framer's thread (headers, CRC-32C, transactions of the next fetch), the pipeline's worker (stage 1 of
                        # up to `depth` fetches ahead) and this one (interning + fold of the oldest push).  The worker asks for
                        # fetch i + depth only after push i is finished: asking tells the framer that fetch i's arena / slab may
                        # be framed into again, and a push reads its bytes until then.
                        with PushPipeline(framed, push, depth) as pipe:
                            for _ in pipe:
                                finish_one()
                                pipe.done()
                    else:
                        pending = 0
                        fetch_iter = iter(framed)
                        while True:
                            # the oldest push is finished BEFORE the next fetch is asked for (see above)
                            if pending == depth:
                                finish_one()
                                pending -= 1
                            item = next(fetch_iter, None)
                            if item is None:
                                break
                            if push(item):
                                pending += 1
                        while pending:
                            finish_one()
                            pending -= 1
                finally:
                    if d is not None:  # (an error path: pushes that are enqueued read the framer's slabs until they are finished)
                        while d.pending:
                            try:
                                d.finish()
                            except Exception:
                                pass
                if d is not None:
                    self.engine.synchronize()
                counters = framed.counters()
            if bound_log and d is not None:
                n_agg = max(n_agg, capacity, d.n_keys)
                self.engine.pack_staged(n_agg)  # decoded fetches -> ONE bound CSR log
                self.engine.fold(algo)
                self.engine.synchronize()
            elif n_agg < 0:  # nothing deliverable in the whole topic
                n_agg = capacity
                self.engine.load_csr(np.zeros(n_agg + 1, dtype=np.int64), np.zeros(0, dtype=EVENT_DTYPE))
                self.engine.fold()
            keys = KeyTable()
            if d is not None:
                for k in d.keys():
                    keys.intern(k)
                counters.update(d.counters())
            self.keys = keys
        finally:
            if d is not None:
                d.close()
        self.engine.snapshot()  # publishes the host mirror that serves point reads
        self._restored = True
        return counters

    def restore_log(self, log: EventLog, init_state: Optional[np.ndarray] = None, algo: int = ALGO_AUTO) -> None:
        self.keys = log.keys
        self.engine.load_csr(log.seg_off, log.events, init_state)
        self.engine.fold(algo)
        self.engine.snapshot()  # publishes the host mirror that serves point reads
        self._restored = True

    def apply_events(self, events_in_offset_order: Sequence) -> None:
        """Streaming micro-batch (config C5): fold new events onto the resident state."""
        if not self._restored:
            raise RuntimeError("restore() first")
        enc = self.model.encode_events(events_in_offset_order)  # may raise: nothing has been interned yet
        ids = [self.model.aggregate_id_of(e) for e in events_in_offset_order]
        # aggregates that first appear after recovery are the normal Surge case: the resident state grows
        # (surge_replay_grow) BEFORE any id is interned, so a failure leaves the key table consistent
        fresh = len({k for k in ids if self.keys.get(k) is None})
        if len(self.keys) + fresh > self.engine.n_agg:
            self.engine.grow(len(self.keys) + fresh)
        agg_idx = np.fromiter((self.keys.intern(k) for k in ids), dtype=np.int64, count=len(ids))
        self.engine.append_events(agg_idx, enc)  # stable group-by + K3 inside the library
        self.engine.snapshot()

    # -- S2 ---------------------------------------------------------------------------------
    def get_aggregate_bytes(self, aggregate_id: str) -> Optional[bytes]:
        """``None`` = no such aggregate (KTable miss) or tombstone; else ``writeState(state).value``."""
        agg = self.get_aggregate(aggregate_id)
        if agg is None:
            return None
        return self.business_logic.aggregate_write_formatting().write_state(agg).value

    def get_aggregate(self, aggregate_id: str):
        idx = self.keys.get(aggregate_id)
        if idx is None or idx >= self.engine.n_agg:
            return None
        raw = self.engine.get_raw(idx)
        if int(raw["flags"]) & STATE_POISONED:
            raise AggregateInitializationException(
                f"replay of aggregate {aggregate_id!r} hit an event whose handler throws; state is frozen before it"
            )
        if not int(raw["flags"]) & STATE_PRESENT:
            return None
        return self.model.state_from_fixed(aggregate_id, raw)


class GpuReplayKeyValueStore:
    """The store a ``GpuReplayPersistencePlugin.create_supplier`` hands to the KTable topology.

    Reads fall through to the GPU-recovered snapshot; ``put`` overlays later state-topic records
    (last write wins, ``None`` is a tombstone).  Read API and semantics follow
    ``KafkaStreamsKeyValueStore`` (``.../streams/KafkaStreamsKeyValueStore.scala:24-54``,
    pinned by ``KafkaStreamsKeyValueStoreSpec.scala:37-91``).
    """

    _TOMBSTONE = object()

    def __init__(self, name: str, recovered: Optional[GpuReplayStateStore] = None):
        self.name = name
        self.recovered = recovered
        self._overlay: Dict[str, object] = {}

    def put(self, key: str, value: Optional[bytes]) -> None:
        self._overlay[key] = self._TOMBSTONE if value is None else bytes(value)

    def delete(self, key: str) -> None:
        self.put(key, None)

    def get(self, key: str) -> Optional[bytes]:
        if key in self._overlay:
            v = self._overlay[key]
            return None if v is self._TOMBSTONE else v
        if self.recovered is not None:
            return self.recovered.get_aggregate_bytes(key)
        return None

    def _all_keys(self) -> List[str]:
        ks = set(self._overlay)
        if self.recovered is not None:
            ks.update(self.recovered.keys.keys)
        return sorted(ks)

    def all(self) -> Iterator[Tuple[str, bytes]]:
        for k in self._all_keys():
            v = self.get(k)
            if v is not None:
                yield k, v

    def all_values(self) -> List[bytes]:
        return [v for _, v in self.all()]

    def range(self, from_key: str, to_key: str) -> Iterator[Tuple[str, bytes]]:
        ks = self._all_keys()
        for k in ks[bisect.bisect_left(ks, from_key): bisect.bisect_right(ks, to_key)]:
            v = self.get(k)
            if v is not None:
                yield k, v

    def approximate_num_entries(self) -> int:
        return sum(1 for _ in self.all())


class GpuReplayPersistencePlugin:
    """``SurgeKafkaStreamsPersistencePlugin`` (…PersistencePlugin.scala:12-15) for the GPU store.

    ``enable_logging`` is False: the store is rebuilt from the events topic, so it needs no
    changelog (the reference then builds the topology un-optimised, SurgeStateStoreConsumer.scala:63-75).
    """

    enable_logging = False

    def __init__(self, recovered: Optional[GpuReplayStateStore] = None):
        self.recovered = recovered

    def create_supplier(self, store_name: str) -> GpuReplayKeyValueStore:
        return GpuReplayKeyValueStore(store_name, self.recovered)


class _NotDeviceDecodable(ValueError):
    """The topic's values are not something the device decoder reads (JSON text, and the model has no template)."""


def _sniff_value_kind(sections, arena_address: int):
    """What the first deliverable record's value looks like — a 16-byte fixed event or JSON text: decides which device
    decoder a topic gets (a topic is written by one plugin: its values are all of one kind; a value that is not fails the
    push loudly)."""
    import ctypes

    def varlong(buf, pos):
        v, shift = 0, 0
        while True:
            b = buf[pos]
            pos += 1
            v |= (b & 0x7F) << shift
            if not b & 0x80:
                break
            shift += 7
        return (v >> 1) ^ -(v & 1), pos

    from . import _native

    for s in sections:
        if int(s["codec"]) & 0xFF == 3:  # still an LZ4 frame (decoded on the GPU): look into it with the library's host decoder
            frame = ctypes.string_at(arena_address + int(s["byte_off"]), int(s["byte_len"]))
            cap = max(1 << 16, 64 * len(frame))
            out = ctypes.create_string_buffer(cap)
            got = _native.load().surge_lz4_frame_decompress(frame, len(frame), out, cap)
            if got < 0:
                return "json"  # too large to peek into, or damaged: the decoder will say which
            buf = out.raw[: min(got, 4096)]
        else:
            buf = ctypes.string_at(arena_address + int(s["byte_off"]), min(int(s["byte_len"]), 4096))
        pos = 0
        try:
            for _ in range(int(s["n_records"])):
                rlen, pos = varlong(buf, pos)  # record length
                end = pos + rlen
                pos += 1                       # attributes
                _, pos = varlong(buf, pos)     # timestampDelta
                _, pos = varlong(buf, pos)     # offsetDelta
                klen, pos = varlong(buf, pos)
                pos += max(klen, 0)
                vlen, pos = varlong(buf, pos)
                if klen == 0 and vlen == 0:    # the producer's flush record: look at the next one
                    pos = end
                    continue
                return "fixed16" if vlen == 16 and buf[pos:pos + 1] != b"{" else "json"
        except IndexError:
            return "json"
    return None
