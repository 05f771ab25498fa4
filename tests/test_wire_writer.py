"""The independent test-side producer (``tests/native/wire_writer.c``) — CPU-only.

Three legs: (1) the C writer == ``tests/kafka_wire.py`` byte for byte (two restatements of the published formats that
share no code with each other or with the product), and its primitives against third parties (RFC 3720's CRC-32C
vectors, the ``xxhash`` module, liblz4 as bundled by Apache Arrow); (2) the topics it writes — one transaction per flush
per partition with COMMIT markers, aborted + retried flushes, markers that arrive a fetch late
(``KafkaProducerActorImpl.scala:397-453``) — read back by the product's host decoder record for record; (3) the
``surge_ingest_group`` behaviour those topics exercise: a marker-only fetch that delivers hundreds of queued batches,
all-or-nothing feeds, host-side LZ4 in a group."""
import ctypes
import random

import numpy as np
import pytest

import kafka_wire as kw
import topic_gen
from surge_amd import _native
from surge_amd.ingest import READ_COMMITTED, READ_UNCOMMITTED, EventsTopicIngest, IngestError, PartitionedFramedFetches


def _arrays(records):
    keys = b"".join(k for k, _ in records)
    vals = b"".join(v for _, v in records)
    ko = np.cumsum([0] + [len(k) for k, _ in records]).astype(np.int64)
    vo = np.cumsum([0] + [len(v) for _, v in records]).astype(np.int64)
    return np.frombuffer(keys + b"\0", np.uint8), ko, np.frombuffer(vals + b"\0", np.uint8), vo


def _c_batch(records, base_offset, flags, pid=-1, epoch=0, seq=-1, ts=0, deltas=None):
    L = topic_gen.wire_lib()
    k, ko, v, vo = _arrays(records)
    out = ctypes.create_string_buffer(4096 + 3 * (len(k) + len(v)) + 64 * len(records))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    d = np.asarray(deltas, np.int64) if deltas is not None else None
    n = L.surge_test_wire_batch(out, base_offset, len(records), None, p(k), p(ko), p(v), p(vo), p(d) if d is not None else None, flags, pid, epoch, seq, ts)
    assert n > 0
    return out.raw[:n]


def test_primitives_against_third_parties():
    L = topic_gen.wire_lib()
    crc = lambda b: L.surge_test_wire_crc32c(b, len(b))  # noqa: E731
    assert crc(b"123456789") == 0xE3069283 and crc(bytes(32)) == 0x8A9136AA and crc(b"\xff" * 32) == 0x62A8AB43  # check value; RFC 3720 B.4
    assert crc(bytes(range(32))) == 0x46DD794E and crc(bytes(range(31, -1, -1))) == 0x113FDB5C
    import xxhash

    rnd = random.Random(1)
    for n in range(0, 16):
        b = bytes(rnd.randrange(256) for _ in range(n))
        for seed in (0, 1, 0x9E3779B1):
            assert L.surge_test_wire_xxh32_short(b, n, seed) == xxhash.xxh32(b, seed=seed).intdigest()


def test_lz4_frames_are_what_liblz4_reads_back():
    """Every shape: empty, below the 13-byte match limit, compressible, incompressible (stored blocks), several blocks, a block
    edge — decoded by liblz4 itself (Apache Arrow's bundled copy) and by the product's host reader."""
    pa = pytest.importorskip("pyarrow")
    L = topic_gen.wire_lib()
    P = _native.load()
    rnd = random.Random(2)
    text = b"".join(b'{"aggregateId":"acct-%08d","incrementBy":%d,"sequenceNumber":%d,"_type":"countIncremented"}' % (rnd.randrange(10 ** 8), rnd.randrange(1000), i) for i in range(3000))
    cases = [b"", b"a", b"abcdefghijkl", b"a" * 13, b"a" * 100000, text, text[:65536], text[:65537], text[:131072 + 5], bytes(rnd.randrange(256) for _ in range(70000)),
             text[:1000] + bytes(rnd.randrange(256) for _ in range(66000)) + text[:3000]]
    for data in cases:
        out = ctypes.create_string_buffer(len(data) + len(data) // 255 + 64 + 8 * (len(data) // 65536 + 1))
        n = L.surge_test_wire_lz4_frame(data, len(data), out)
        frame = out.raw[:n]
        assert frame[:7] == kw.lz4_frame(b"")[:7]  # magic, FLG, BD, HC
        if data:
            assert pa.Codec("lz4").decompress(frame, decompressed_size=len(data)).to_pybytes() == data
        dst = ctypes.create_string_buffer(len(data) + 16)
        assert P.surge_lz4_frame_decompress(frame, n, dst, len(data) + 16) == len(data) and dst.raw[: len(data)] == data
    assert len(frame) < len(cases[-1]) + 64 and n < len(text)  # (stored blocks cost 4 bytes each; text does compress)


def test_batches_and_markers_equal_the_python_writer_byte_for_byte():
    rnd = random.Random(3)
    for trial in range(40):
        n = rnd.choice([1, 2, 7, 130, 300])
        records = [(b"acct-%08d:%d" % (rnd.randrange(10 ** 8), i + 1), bytes(rnd.randrange(32, 127) for _ in range(rnd.choice([0, 1, 63, 64, 110, 200])))) for i in range(n)]
        txn = rnd.random() < 0.5
        pid, epoch, seq = (rnd.randrange(1 << 40), rnd.randrange(5), rnd.randrange(1 << 20)) if txn else (-1, 0, -1)
        base, ts = rnd.randrange(1 << 45), 1700000000000 + rnd.randrange(10 ** 6)
        deltas = sorted(rnd.randrange(0, rnd.choice([1, 50, 5000])) for _ in range(n)) if trial % 2 else None
        got = _c_batch(records, base, topic_gen.WIRE_TRANSACTIONAL if txn else 0, pid, epoch, seq, ts, deltas)
        want = kw.record_batch(base, records, transactional=txn, producer_id=pid, producer_epoch=epoch, base_sequence=seq, base_timestamp=ts, timestamp_deltas=deltas)
        assert got == want
        # lz4: the two compressors differ; the header fields and the decompressed records section do not
        got_z = _c_batch(records, base, topic_gen.WIRE_LZ4 | (topic_gen.WIRE_TRANSACTIONAL if txn else 0), pid, epoch, seq, ts, deltas)
        want_z = kw.record_batch(base, records, compression="lz4", transactional=txn, producer_id=pid, producer_epoch=epoch, base_sequence=seq, base_timestamp=ts, timestamp_deltas=deltas)
        assert got_z[:17] != want_z[:17] or got_z == want_z  # (length / CRC differ when the frames do)
        assert got_z[21:61] == want_z[21:61]
        pa = pytest.importorskip("pyarrow")
        assert pa.Codec("lz4").decompress(got_z[61:], decompressed_size=len(want) - 61).to_pybytes() == want[61:]
    L = topic_gen.wire_lib()
    out = ctypes.create_string_buffer(128)
    for kind in (kw.ABORT, kw.COMMIT):
        n = L.surge_test_wire_control(out, 12345, 77, 3, kind, 1700000000123)
        assert n == 78 and out.raw[:n] == kw.control_batch(12345, 77, kind, producer_epoch=3, timestamp=1700000000123)


def _topic(P, n, rnd):
    part = np.array([rnd.randrange(P) for _ in range(n)], np.int32)
    records = [(b"acct-%08d:%d" % (rnd.randrange(500), i), b'{"aggregateId":"x","incrementBy":%d,"sequenceNumber":%d,"_type":"countIncremented"}' % (rnd.randrange(1000), i)) for i in range(n)]
    return part, records


@pytest.mark.parametrize("codec", ["lz4", "none"])
def test_transactional_topic_reads_back_record_for_record_through_the_host_decoder(codec):
    """K records per flush, a COMMIT per flush, every third flush aborted and retried, markers held back across fetches on
    the partitions p % 4 == 1: read_committed delivers every record exactly once, in order; read_uncommitted also sees the
    aborted copies; the counters name what was on the wire."""
    rnd = random.Random(4)
    P = 6
    with topic_gen.WireTopic(P, flush_events=37, max_batch_bytes=2048, codec=codec, abort_every=3, hold_markers=4) as topic:
        fetches, expect = [], [[] for _ in range(P)]
        for f in range(5):
            part, records = _topic(P, 700 + 50 * f, rnd)
            k, ko, v, vo = _arrays(records)
            fetches.append(topic.fetch(part, k, ko, v, vo))
            for p, r in zip(part, records):
                expect[p].append(r)
        held = topic.fetch(np.zeros(0, np.int32), None, None, None, None)
        assert [bool(x) for x in held] == [p % 4 == 1 for p in range(P)] and all(len(x) == 78 for x in held if x)
        fetches.append(held)
        counts = topic.counts
        ends = topic.end_offsets()
    n_rec = sum(len(e) for e in expect)
    assert counts["transactions"] == counts["control_batches"] and counts["records_written"] == n_rec + counts["records_aborted"] and counts["records_aborted"] > 0
    assert sum(ends) == counts["records_written"] + counts["control_batches"]
    for level, extra in ((READ_COMMITTED, 0), (READ_UNCOMMITTED, counts["records_aborted"])):
        got_total = 0
        for p in range(P):
            with EventsTopicIngest(isolation_level=level) as g:
                got = []
                for f in fetches:
                    if f[p]:
                        g.feed(f[p])
                        got += [(k, v) for _, _, k, v in g.drain_records()]
                c = g.counters()
            if level == READ_COMMITTED:
                assert got == expect[p]
                assert c["open_transactions"] == 0
            got_total += len(got)
        assert got_total == n_rec + extra


def test_group_delivers_hundreds_of_queued_batches_when_only_the_marker_arrives():
    """400 transactional batches in fetch 1, their COMMIT marker alone in fetch 2 — the second feed's
    section table must be sized from what the group holds, not from the 78 bytes it is fed."""
    recs = [(b"k%d:1" % i, b"v" * 40) for i in range(3)]
    batches = [kw.record_batch(3 * i, recs, transactional=True, producer_id=9) for i in range(400)]
    f1 = [b"".join(batches), None]
    f2 = [kw.control_batch(1200, 9, kw.COMMIT), None]
    with PartitionedFramedFetches(iter([f1, f2]), 2, threads=2, hold=2, overlap=False) as framed:
        got = [sec.shape[0] for sec, _ in framed]
        assert got == [0, 400] and framed.counters()["records_delivered"] == 1200


def test_group_feed_is_all_or_nothing():
    """A corrupt batch in one partition, or a section table that is too small: the feed reports it and EVERY partition's
    framer is what it was — feeding the repaired response afterwards delivers exactly what an undisturbed group delivers."""
    L = _native.load()
    from surge_amd.ingest import SECTION_DTYPE

    rnd = random.Random(5)
    P = 4
    with topic_gen.WireTopic(P, flush_events=20, max_batch_bytes=1024, codec="lz4", abort_every=4, hold_markers=2) as topic:
        fetches = []
        for f in range(4):
            part, records = _topic(P, 400, rnd)
            fetches.append(topic.fetch(part, *_arrays(records)))
        fetches.append(topic.fetch(np.zeros(0, np.int32), None, None, None, None))

    def run(disturb):
        h = ctypes.c_void_p()
        assert L.surge_ingest_group_create(P, READ_COMMITTED | 0x200, ctypes.byref(h)) == 0  # DEVICE_LZ4
        out = []
        try:
            for i, fetch in enumerate(fetches):
                bufs = [x or b"" for x in fetch]
                secs = np.zeros(4096, SECTION_DTYPE)
                n_sec, slab = ctypes.c_int64(), ctypes.c_void_p()
                consumed = (ctypes.c_int64 * P)()

                def feed(b, cap):
                    arr = (ctypes.c_void_p * P)(*[ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p) if x else None for x in b])
                    ln = (ctypes.c_int64 * P)(*[len(x) for x in b])
                    return L.surge_ingest_group_feed(h, arr, ln, 3, consumed, cap, secs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n_sec), ctypes.byref(slab))

                if disturb and i in (1, 3):
                    bad = list(bufs)
                    j = len(bad[2]) // 2
                    bad[2] = bad[2][:j] + bytes([bad[2][j] ^ 0x55]) + bad[2][j + 1:]  # partition 2: a CRC failure somewhere in the middle
                    assert feed(bad, 4096) == -7 and b"partition 2" in L.surge_ingest_group_last_error(h) and list(consumed) == [0] * P and n_sec.value == 0
                    assert feed(bufs, 1) == -1 and n_sec.value > 1 and list(consumed) == [0] * P  # too small: says what it needs
                    assert b"too small" in L.surge_ingest_group_last_error(h)
                assert feed(bufs, 4096) == 0 and list(consumed) == [len(x) for x in bufs]
                rows = secs[: n_sec.value]
                out.append([(int(s["base_offset"]), int(s["n_records"]), int(s["codec"]), ctypes.string_at(slab.value + int(s["byte_off"]), int(s["byte_len"]))) for s in rows])
            c = (ctypes.c_int64 * 8)()
            L.surge_ingest_group_counters(h, ctypes.byref(c))
            return out, list(c)
        finally:
            L.surge_ingest_group_destroy(h)

    clean, disturbed = run(False), run(True)
    assert clean == disturbed and clean[1][2] == 1600 and clean[1][3] > 0


def test_group_sizes_all_its_slabs_at_the_first_feed():
    """Round 5: page-locking a slab costs milliseconds (sometimes tens), so the group's FIRST feed allocates all six slabs, with a
    quarter to spare — later fetch responses of the same size, and somewhat larger ones, allocate nothing; a response that
    outgrows the spare re-sizes only the slab it is framed into.  Counted through the pluggable allocator the pinned slabs use."""
    L = _native.load()
    from surge_amd.ingest import SECTION_DTYPE

    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = ctypes.c_void_p, [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    sizes, freed = [], []
    ALLOC, FREE = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t), ctypes.CFUNCTYPE(None, ctypes.c_void_p)

    @ALLOC
    def alloc(n):
        sizes.append(n)
        return libc.malloc(n)

    @FREE
    def release(ptr):
        freed.append(ptr)
        libc.free(ptr)

    P = 3
    rnd = random.Random(17)
    h = ctypes.c_void_p()
    assert L.surge_ingest_group_create(P, READ_COMMITTED | 0x200, ctypes.byref(h)) == 0
    try:
        L.surge_ingest_group_set_allocator.argtypes = [ctypes.c_void_p, ALLOC, FREE]
        assert L.surge_ingest_group_set_allocator(h, alloc, release) == 0
        off = [0] * P

        def feed(n_rec, val_len):
            bufs = []
            for p in range(P):
                recs = [(b"k%d:%d" % (p, i), rnd.randbytes(val_len)) for i in range(n_rec)]  # (incompressible: the frames are as large as the records)
                bufs.append(kw.record_batch(off[p], recs, compression="lz4"))
                off[p] += n_rec
            arr = (ctypes.c_void_p * P)(*[ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p) for x in bufs])
            ln = (ctypes.c_int64 * P)(*[len(x) for x in bufs])
            secs = np.zeros(64, SECTION_DTYPE)
            n_sec, slab = ctypes.c_int64(), ctypes.c_void_p()
            consumed = (ctypes.c_int64 * P)()
            assert L.surge_ingest_group_feed(h, arr, ln, 2, consumed, 64, secs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n_sec), ctypes.byref(slab)) == 0
            assert n_sec.value == P and [ctypes.string_at(slab.value + int(s["byte_off"]), 4) for s in secs[:P]] == [b"\x04\x22\x4d\x18"] * P  # the LZ4 frames, in the slab
            return sum(len(x) for x in bufs)

        first = feed(400, 500)
        assert len(sizes) == 6 and len(set(sizes)) == 1 and sizes[0] >= first + first // 4 and not freed  # six slabs, alike, with room to spare
        for _ in range(8):  # (more feeds than slabs: every slab is written again)
            feed(400, 500 + rnd.randrange(50))
        assert len(sizes) == 6 and not freed
        feed(400, 4 * 500)  # a response several times the first: only the slab it goes into is re-sized
        assert len(sizes) == 7 and sizes[6] > sizes[0] and len(freed) == 1
    finally:
        L.surge_ingest_group_destroy(h)
    assert len(freed) == 7  # everything the group allocated went back through the allocator


def test_group_without_device_lz4_decompresses_into_slices_it_sizes_by_trial():
    """host-side LZ4 in a group — a batch that expands 20 x does not fit a slice sized for its compressed
    bytes; the group undoes the feed and runs it again with more room instead of failing with 'out of host memory'."""
    big = [(b"k:%d" % i, b"a" * 2000) for i in range(40)]  # 80 KB of records in a ~1 KB lz4 batch
    small = [(b"j:%d" % i, b"b" * 10) for i in range(5)]
    f1 = [kw.record_batch(0, big, compression="lz4"), kw.record_batch(0, small, compression="lz4")]
    plain = [kw.record_batch(0, big)[61:], kw.record_batch(0, small)[61:]]
    with PartitionedFramedFetches(iter([f1]), 2, threads=2, hold=1, overlap=False, device_lz4=False) as framed:
        (sec, slab), = list(framed)
        assert [int(s["codec"]) for s in sec] == [0, 0]
        assert [ctypes.string_at(slab + int(s["byte_off"]), int(s["byte_len"])) for s in sec] == plain


def test_partitioned_fetches_reject_a_short_fetch_response_before_feeding_anything():
    ok = [kw.record_batch(0, [(b"a:1", b"x")]), kw.record_batch(0, [(b"b:1", b"y")])]
    with PartitionedFramedFetches(iter([ok[:1], ok]), 2, threads=1, hold=1, overlap=False) as framed:
        it = iter(framed)
        with pytest.raises(ValueError, match="2 partitions"):
            next(it)
        sec, _ = next(it)  # the framer is intact: the complete response frames
        assert sec.shape[0] == 2
