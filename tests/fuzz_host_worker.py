"""TEST INFRASTRUCTURE: mutation fuzzing of the host-side decoders / encoders (ingest.cpp, lz4_frame.cpp,
snapshot_writer.cpp) built with AddressSanitizer + UBSan.  Run as a subprocess with LD_PRELOAD=libasan.so by
tests/test_host_fuzz.py:  python fuzz_host_worker.py <libsurge_host_asan.so> <seconds> <seed>

Bytes from a Kafka fetch are untrusted input: whatever they are, the decoder must answer with a status code — never read or
write outside its buffers.  A sanitizer report aborts this process; the parent test shows it."""
import ctypes
import os
import random
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kafka_wire as kw  # noqa: E402

lib = ctypes.CDLL(sys.argv[1])
budget, seed = float(sys.argv[2]), int(sys.argv[3])
rng = random.Random(seed)
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.surge_ingest_create.argtypes = [i32, ctypes.POINTER(vp)]
lib.surge_ingest_destroy.argtypes = [vp]
lib.surge_ingest_feed.argtypes = [vp, ctypes.c_char_p, i64, ctypes.POINTER(i64)]
lib.surge_ingest_ready.argtypes = [vp]
lib.surge_ingest_ready.restype = i64
lib.surge_ingest_drain_fixed16.argtypes = [vp, i64, vp, vp, vp, ctypes.POINTER(i64)]
lib.surge_ingest_drain.argtypes = [vp, i64, vp, ctypes.POINTER(i64)]
lib.surge_ingest_arena.argtypes = [vp]
lib.surge_ingest_arena.restype = ctypes.POINTER(ctypes.c_uint8)


class Rec(ctypes.Structure):  # surge_ingest_record
    _fields_ = [("offset", i64), ("agg_idx", i64), ("key_off", i64), ("key_len", i32), ("value_len", i32), ("value_off", i64)]
lib.surge_lz4_frame_decompress.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64]
lib.surge_lz4_frame_decompress.restype = i64
lib.surge_lz4_frame_bound.argtypes = [i64]
lib.surge_lz4_frame_bound.restype = i64
lib.surge_lz4_frame_compress.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64]
lib.surge_lz4_frame_compress.restype = i64


class EvType(ctypes.Structure):  # surge_event_json_type
    _fields_ = [("name", ctypes.c_char * 64), ("event_type", ctypes.c_uint32), ("arg_kind", ctypes.c_uint32), ("seq_field", ctypes.c_char * 64),
                ("arg_field", ctypes.c_char * 64)]


class EvTemplate(ctypes.Structure):  # surge_event_json_template
    _fields_ = [("n_types", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("discriminator", ctypes.c_char * 64), ("types", EvType * 16)]


TMPL = EvTemplate()
TMPL.n_types, TMPL.discriminator = 3, b"_type"
for _i, (_n, _t, _k, _s, _a) in enumerate([(b"countIncremented", 1, 1, b"sequenceNumber", b"incrementBy"), (b"no-op", 0, 0, b"sequenceNumber", b""),
                                           (b"upd", 4, 2, b"", b"newBalance")]):
    TMPL.types[_i].name, TMPL.types[_i].event_type, TMPL.types[_i].arg_kind, TMPL.types[_i].seq_field, TMPL.types[_i].arg_field = _n, _t, _k, _s, _a
lib.surge_event_json_decode.argtypes = [vp, ctypes.c_char_p, i64, vp]
lib.surge_parse_f64_json.argtypes = [ctypes.c_char_p, i64, ctypes.POINTER(ctypes.c_uint64)]
lib.surge_format_f64_json.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int32]
lib.surge_ingest_drain_sections.argtypes = [vp, i64, vp, ctypes.POINTER(i64)]
lib.surge_ingest_group_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
lib.surge_ingest_group_destroy.argtypes = [vp]
lib.surge_ingest_group_queued_sections.argtypes = [vp]
lib.surge_ingest_group_queued_sections.restype = i64
lib.surge_ingest_group_receive_copy.argtypes = [vp, vp, vp, i32, vp]
lib.surge_ingest_group_feed.argtypes = [vp, vp, vp, i32, vp, i64, vp, ctypes.POINTER(i64), ctypes.POINTER(vp)]


def valid_wire():
    batches, off = [], rng.randrange(0, 1 << 40)
    for _ in range(rng.randrange(1, 5)):
        n = rng.randrange(1, 30)
        recs = []
        for i in range(n):
            key = None if rng.random() < 0.05 else f"agg-{rng.randrange(40)}:{i}".encode()[: rng.randrange(0, 20)]
            val = None if rng.random() < 0.05 else os.urandom(rng.choice([0, 1, 15, 16, 17, 64, 300]))
            hdrs = [(b"h" * rng.randrange(0, 5), None if rng.random() < 0.3 else b"v" * rng.randrange(0, 9))] if rng.random() < 0.3 else []
            recs.append((key, val, hdrs))
        tx = rng.random() < 0.3
        batches.append(kw.record_batch(off, recs, compression=rng.choice(["none", "lz4"]), transactional=tx, producer_id=7 if tx else -1))
        off += n
        if tx and rng.random() < 0.7:
            batches.append(kw.control_batch(off, 7, rng.choice([kw.COMMIT, kw.ABORT])))
            off += 1
    return b"".join(batches)


def mutate(b: bytes) -> bytes:
    b = bytearray(b)
    for _ in range(rng.randrange(1, 6)):
        if not b:
            break
        kind = rng.randrange(7)
        i = rng.randrange(len(b))
        if kind == 0:
            b[i] ^= 1 << rng.randrange(8)
        elif kind == 1:
            b[i] = rng.choice([0, 0x7F, 0x80, 0xFF])
        elif kind == 2:
            del b[i:i + rng.randrange(1, 40)]
        elif kind == 3:
            b[i:i] = os.urandom(rng.randrange(1, 20))
        elif kind == 4:
            b = b[: rng.randrange(len(b))]
        elif kind == 5 and len(b) >= 4:
            j = rng.randrange(len(b) - 3)
            b[j:j + 4] = struct.pack(rng.choice(["<I", ">I"]), rng.choice([0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, len(b)]))
        else:
            j = rng.randrange(len(b))
            b[i:i] = b[j:j + rng.randrange(1, 64)]
    return bytes(b)


def fix_crc(b: bytes) -> bytes:
    """Re-seal the first batch so mutations get past the CRC gate and reach the record parser / LZ4 decoder."""
    if len(b) < 61:
        return b
    (blen,) = struct.unpack(">i", b[8:12])
    end = 12 + blen
    if blen < 49 or end > len(b):
        return b
    return b[:17] + struct.pack(">I", kw.crc32c(b[21:end])) + b[21:]


def drain(h):
    n = lib.surge_ingest_ready(h)
    if n <= 0:
        return
    if rng.random() < 0.5:
        idx, ev, off = (ctypes.c_int64 * n)(), (ctypes.c_uint8 * (16 * n))(), (ctypes.c_int64 * n)()
        got = i64()
        lib.surge_ingest_drain_fixed16(h, n, idx, ev, off, ctypes.byref(got))
    else:
        recs = (Rec * n)()
        got = i64()
        assert lib.surge_ingest_drain(h, n, recs, ctypes.byref(got)) == 0 and got.value <= n
        arena = lib.surge_ingest_arena(h)
        for r in recs[: got.value]:  # every span the decoder hands out must be readable (ASan checks the reads)
            if r.key_len > 0:
                _ = bytes(arena[r.key_off: r.key_off + r.key_len])
            if r.value_len > 0:
                _ = arena[r.value_off], arena[r.value_off + r.value_len - 1]


t_end, rounds = time.time() + budget, 0
while time.time() < t_end:
    rounds += 1
    wire = valid_wire()
    data = wire if rng.random() < 0.15 else mutate(wire)
    if rng.random() < 0.6:
        data = fix_crc(data)
    h = vp()
    assert lib.surge_ingest_create(rng.randrange(2), ctypes.byref(h)) == 0
    pos = 0
    while pos < len(data):  # arbitrary fetch boundaries; a failing feed ends this stream
        step = rng.randrange(1, 400)
        chunk = data[pos:pos + step]
        consumed = i64()
        rc = lib.surge_ingest_feed(h, chunk, len(chunk), ctypes.byref(consumed))
        assert 0 <= consumed.value <= len(chunk), (consumed.value, len(chunk))
        if rc != 0:
            break
        pos += step
        if rng.random() < 0.5:
            drain(h)
    drain(h)
    lib.surge_ingest_destroy(h)
    # LZ4 frames: liblz4-shaped ones from the product's own writer, mutated
    raw = os.urandom(rng.randrange(0, 300)) + bytes(rng.randrange(0, 70000)) + b"abc" * rng.randrange(0, 3000)
    cap = lib.surge_lz4_frame_bound(len(raw))
    dst = ctypes.create_string_buffer(cap)
    n = lib.surge_lz4_frame_compress(raw, len(raw), dst, cap)
    assert n > 0
    frame = dst.raw[:n]
    out = ctypes.create_string_buffer(len(raw) + 16)
    assert lib.surge_lz4_frame_decompress(frame, n, out, len(raw) + 16) == len(raw) and out.raw[: len(raw)] == raw
    bad = mutate(frame)
    small = rng.choice([0, 1, 100, len(raw) // 2 + 1, len(raw) + 16])
    out2 = ctypes.create_string_buffer(max(small, 1))
    r = lib.surge_lz4_frame_decompress(bad, len(bad), out2, small)
    assert r <= small
    # JSON event values: valid ones must decode, mutated ones must answer 0 or CORRUPT without touching foreign memory
    for _ in range(20):
        good = rng.choice([
            b'{"aggregateId":"a","incrementBy":%d,"sequenceNumber":%d,"_type":"countIncremented"}' % (rng.randrange(-2**31, 2**31), rng.randrange(2**31)),
            b'{"_type":"no-op","aggregateId":"x\\"y","sequenceNumber":7,"n":{"a":[1,{"b":"}"}]}}',
            b'{"_type":"upd","newBalance":%s}' % repr(rng.uniform(-1e300, 1e300)).encode(),
        ])
        ev = (ctypes.c_uint8 * 16)()
        assert lib.surge_event_json_decode(ctypes.byref(TMPL), good, len(good), ev) == 0, good
        m = mutate(good)
        exact = ctypes.create_string_buffer(m, len(m))  # no NUL terminator behind the value: an over-read is an ASan report
        assert lib.surge_event_json_decode(ctypes.byref(TMPL), exact, len(m), ev) in (0, -7)
    # numbers: text -> double (Eisel-Lemire + strtod) and double -> play-json text (Ryu), exact-size buffers
    for _ in range(40):
        num = rng.choice([repr(rng.uniform(-1e300, 1e300)), "%de%d" % (rng.randrange(10 ** 19), rng.randrange(-400, 400)),
                          "0." + "0" * rng.randrange(0, 30) + str(rng.randrange(10 ** 25)), "1e-400", "-0.0"]).encode()
        bits = ctypes.c_uint64()
        assert lib.surge_parse_f64_json(ctypes.create_string_buffer(num, len(num)), len(num), ctypes.byref(bits)) in (0, 1), num
        m = mutate(num)
        assert lib.surge_parse_f64_json(ctypes.create_string_buffer(m, len(m)), len(m), ctypes.byref(bits)) in (0, 1, -7)
        out = ctypes.create_string_buffer(26)
        n = lib.surge_format_f64_json(ctypes.c_uint64(rng.getrandbits(64)), out, 26)
        assert 0 <= n <= 26
    # the framing mode of the decoder (device decode): sections must lie inside the arena, whatever was fed
    # Several fetches through one framer: what a drain handed out must stay readable, unchanged, THROUGH the next feed (the
    # two arenas alternate; transactions left open by one fetch travel to the next arena) — a freed or overwritten arena is
    # an ASan report / a changed byte here.
    hf = vp()
    assert lib.surge_ingest_create(1 | 0x100, ctypes.byref(hf)) == 0
    held = []  # (address, bytes) of the sections of the previous drain
    for fetch in range(rng.randrange(1, 6)):
        wire2 = mutate(valid_wire()) if rng.random() < 0.5 else valid_wire()
        consumed2 = i64()
        lib.surge_ingest_feed(hf, wire2, len(wire2), ctypes.byref(consumed2))
        for addr, data in held:
            assert ctypes.string_at(addr, len(data)) == data
        if rng.random() < 0.25:
            continue  # a feed that is followed by no drain: the next feed keeps appending (and may grow) the same arena
        secs = (i64 * (4 * 64))()
        n_sec = i64()
        assert lib.surge_ingest_drain_sections(hf, 64, secs, ctypes.byref(n_sec)) == 0
        if n_sec.value:
            held = []
        for k in range(n_sec.value):
            byte_off, byte_len = secs[4 * k], secs[4 * k + 1]
            assert byte_off >= 0 and byte_len >= 0
            if byte_len:
                addr = ctypes.addressof(lib.surge_ingest_arena(hf).contents) + byte_off
                held.append((addr, ctypes.string_at(addr, byte_len)))  # readable end to end (ASan checks)
    lib.surge_ingest_destroy(hf)
    # The partition group (surge_ingest_group_feed): a framing thread walks the batch LENGTHS of eight partitions side by
    # side before it frames them (round 6) — lengths read out of whatever the responses hold.  9 .. 17 partitions of valid,
    # mutated, truncated and empty responses, each in a buffer of exactly its size (a read past its end is an ASan report),
    # framed by copy and in place (received into the group's slab first); every section a successful feed hands out lies
    # inside the slab.
    P = rng.randrange(9, 18)
    dev_lz4 = rng.random() < 0.7
    flags = rng.randrange(2) | (0x200 if dev_lz4 else 0) | (0x400 if dev_lz4 and rng.random() < 0.6 else 0)  # isolation | SURGE_INGEST_DEVICE_LZ4 | _DEVICE_CRC
    grp = vp()
    assert lib.surge_ingest_group_create(P, flags, ctypes.byref(grp)) == 0
    for fetch in range(rng.randrange(1, 4)):
        bufs = []
        for q in range(P):
            w = valid_wire()
            r = rng.random()
            w = b"" if r < 0.1 else w[: rng.randrange(0, 80)] if r < 0.25 else fix_crc(mutate(w)) if r < 0.5 else mutate(w) if r < 0.6 else w
            bufs.append(ctypes.create_string_buffer(w, len(w)) if w else None)
        lens = (i64 * P)(*[len(b) if b is not None else 0 for b in bufs])
        src = (vp * P)(*[ctypes.addressof(b) if b is not None else None for b in bufs])
        data = src
        if dev_lz4 and rng.random() < 0.7:  # in place
            placed = (vp * P)()
            if lib.surge_ingest_group_receive_copy(grp, src, lens, rng.randrange(1, 4), placed) != 0:
                break
            data = placed
        cap = sum(lens) // 61 + int(lib.surge_ingest_group_queued_sections(grp)) + 16
        secs = (i64 * (4 * cap))()
        n_sec, slab = i64(), vp()
        consumed = (i64 * P)()
        rc = lib.surge_ingest_group_feed(grp, data, lens, rng.randrange(1, 4), consumed, cap, secs, ctypes.byref(n_sec), ctypes.byref(slab))
        if rc != 0:
            continue  # undone: the group is what it was
        for q in range(P):
            assert 0 <= consumed[q] <= lens[q]
        for k in range(n_sec.value):
            byte_off, byte_len = secs[4 * k], secs[4 * k + 1]
            assert byte_off >= 0 and byte_len >= 0
            if byte_len:
                _ = ctypes.string_at((slab.value or 0) + byte_off, byte_len)  # readable end to end (ASan checks)
    lib.surge_ingest_group_destroy(grp)
print(f"OK {rounds} rounds")
