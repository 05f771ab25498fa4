"""TEST INFRASTRUCTURE: a writer for Kafka message-format-v2 record batches and LZ4 frames.

No Kafka client exists in this image, so the producer side is restated here from the published formats
(KIP-98 record batch v2; LZ4 frame format 1.6) independently of the C++ decoder in
``surge_amd/csrc/ingest.cpp`` — two implementations that must agree.  Shape of what the reference's
producer writes: ``compression.type=lz4``, idempotent transactional producer, one transaction per flush
(``modules/common/src/main/resources/reference.conf:111-126``,
``KafkaProducerActorImpl.scala:421-453``).
"""
import struct

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = (c >> 8) ^ _CRC_TABLE[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def varint(v: int) -> bytes:
    """Zig-zag varint / varlong (ByteUtils.writeVarlong)."""
    z = (v << 1) ^ (v >> 63)
    z &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = z & 0x7F
        z >>= 7
        if z:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def record(offset_delta: int, key, value, headers=(), timestamp_delta: int = 0) -> bytes:
    body = bytearray()
    body += b"\x00"  # attributes
    body += varint(timestamp_delta)
    body += varint(offset_delta)
    if key is None:
        body += varint(-1)
    else:
        body += varint(len(key)) + key
    if value is None:
        body += varint(-1)
    else:
        body += varint(len(value)) + value
    body += varint(len(headers))
    for hk, hv in headers:
        body += varint(len(hk)) + hk
        body += varint(-1) if hv is None else varint(len(hv)) + hv
    return varint(len(body)) + bytes(body)


def lz4_block_compress(data: bytes) -> bytes:
    """A small greedy LZ4 block compressor (4-byte hash matches, offsets < 64 KiB)."""
    n = len(data)
    out = bytearray()
    table = {}
    anchor = 0
    i = 0
    last_literals = 5  # LZ4 end conditions: the last 5 bytes are literals, matches end 12 bytes before the end
    while i + 4 <= n - 12 if n >= 13 else False:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 0xFFFF:
            ml = 4
            while i + ml < n - last_literals and data[cand + ml] == data[i + ml]:
                ml += 1
            lit = i - anchor
            token_l = min(lit, 15)
            token_m = min(ml - 4, 15)
            out.append((token_l << 4) | token_m)
            if lit >= 15:
                r = lit - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            out += data[anchor:i]
            out += struct.pack("<H", i - cand)
            if ml - 4 >= 15:
                r = ml - 4 - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            i += ml
            anchor = i
        else:
            i += 1
    lit = n - anchor
    out.append(min(lit, 15) << 4)
    if lit >= 15:
        r = lit - 15
        while r >= 255:
            out.append(255)
            r -= 255
        out.append(r)
    out += data[anchor:]
    return bytes(out)


def header_checksum(descriptor: bytes) -> int:
    """HC of an LZ4 frame: second byte of XXH32(descriptor, seed 0) — from the third-party xxhash module."""
    import xxhash

    return (xxhash.xxh32(descriptor, seed=0).intdigest() >> 8) & 0xFF


def lz4_frame(data: bytes, block_size: int = 65536, store_incompressible: bool = True, content_checksum: bool = False) -> bytes:
    """LZ4 frame: magic, FLG (version 01, block independence), BD (64 KiB), HC, blocks, EndMark."""
    flg = 0x60 | (0x04 if content_checksum else 0)
    out = bytearray(struct.pack("<I", 0x184D2204))
    out += bytes([flg, 0x40, header_checksum(bytes([flg, 0x40]))])
    for s in range(0, len(data), block_size):
        chunk = data[s:s + block_size]
        comp = lz4_block_compress(chunk)
        if store_incompressible and len(comp) >= len(chunk):
            out += struct.pack("<I", len(chunk) | 0x80000000) + chunk
        else:
            out += struct.pack("<I", len(comp)) + comp
    out += struct.pack("<I", 0)
    if content_checksum:
        out += b"\x00\x00\x00\x00"
    return bytes(out)


COMMIT, ABORT = 1, 0


def record_batch(base_offset: int, records, *, compression: str = "none", transactional: bool = False, control: bool = False,
                 producer_id: int = -1, producer_epoch: int = 0, base_sequence: int = -1, base_timestamp: int = 0,
                 magic: int = 2, codec_override=None, compressor=None, timestamp_deltas=None) -> bytes:
    """``records``: list of (key, value) or (key, value, headers); offsets are base_offset + index.  ``compressor``
    replaces this module's own LZ4 frame writer (e.g. with the reference lz4 library as bundled by Apache Arrow).
    ``timestamp_deltas``: per record, ms after ``base_timestamp`` (maxTimestamp follows)."""
    td = list(timestamp_deltas) if timestamp_deltas is not None else [0] * len(records)
    recs = b"".join(record(i, *(r if len(r) == 3 else (r[0], r[1], ())), timestamp_delta=td[i]) for i, r in enumerate(records))
    codec = {"none": 0, "gzip": 1, "snappy": 2, "lz4": 3, "zstd": 4}[compression] if codec_override is None else codec_override
    payload = (compressor(recs) if compressor else lz4_frame(recs)) if compression == "lz4" else recs
    attrs = codec | (0x10 if transactional else 0) | (0x20 if control else 0)
    n = len(records)
    after_crc = struct.pack(">hiqqqhii", attrs, max(n - 1, 0), base_timestamp, base_timestamp + max(td, default=0), producer_id, producer_epoch,
                            base_sequence, n) + payload
    body = struct.pack(">ib", 0, magic) + struct.pack(">I", crc32c(after_crc)) + after_crc
    return struct.pack(">qi", base_offset, len(body)) + body


def control_batch(offset: int, producer_id: int, kind: int, producer_epoch: int = 0, timestamp: int = 0) -> bytes:
    key = struct.pack(">hh", 0, kind)         # version, type (0 ABORT, 1 COMMIT)
    value = struct.pack(">hi", 0, 0)          # version, coordinatorEpoch
    return record_batch(offset, [(key, value)], transactional=True, control=True, producer_id=producer_id,
                        producer_epoch=producer_epoch, base_timestamp=timestamp)
