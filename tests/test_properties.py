"""Property tests (CPU): the oracle's fold is a left fold — splitting a log anywhere and resuming from the
intermediate snapshot gives the same bytes; and the decoder never crashes on damaged input."""
import os
import random

import numpy as np
from hypothesis import given, settings, strategies as st

import kafka_wire as kw
from oracle import oracle
from surge_amd import schema as S
from surge_amd.ingest import EventsTopicIngest, IngestError

events_strategy = st.lists(
    st.tuples(st.integers(0, 8), st.integers(0, 2**31 - 1), st.integers(-(2**31), 2**31 - 1)), min_size=0, max_size=60)


@settings(max_examples=200, deadline=None)
@given(events_strategy, st.integers(0, 60))
def test_fold_resumes_from_any_intermediate_snapshot(evs, cut):
    ev = S.make_events([e[0] for e in evs], [e[1] for e in evs], [e[2] for e in evs])
    cut = min(cut, len(evs))
    whole = oracle.fold_csr(np.array([0, len(evs)], dtype=np.int64), ev)
    first = oracle.fold_csr(np.array([0, cut], dtype=np.int64), ev[:cut])
    second = oracle.fold_csr(np.array([0, len(evs) - cut], dtype=np.int64), ev[cut:], first)
    assert second.tobytes() == whole.tobytes()


@settings(max_examples=100, deadline=None)
@given(events_strategy)
def test_poison_is_sticky_and_none_is_canonical(evs):
    ev = S.make_events([e[0] for e in evs], [e[1] for e in evs], [e[2] for e in evs])
    out = oracle.fold_csr(np.array([0, len(evs)], dtype=np.int64), ev)[0]
    fl = int(out["flags"])
    if not fl & S.STATE_PRESENT:
        zero = S.empty_states(1)[0].copy()
        zero["flags"] = fl
        assert out.tobytes() == zero.tobytes()  # None is all-zero apart from the poison bit
    if fl & S.STATE_POISONED:
        # appending anything after a throwing event changes nothing
        more = S.make_events([S.EVT_INC, S.EVT_DELETE], [1, 2], [5, 0])
        again = oracle.fold_csr(np.array([0, 2], dtype=np.int64), more, np.array([out]))
        assert again[0].tobytes() == out.tobytes()


def test_damaged_record_batches_never_crash_the_decoder():
    rng = random.Random(0)
    good = b"".join([
        kw.record_batch(0, [(b"a:1", os.urandom(16)), (b"b:1", os.urandom(16))], compression="lz4"),
        kw.record_batch(2, [(b"a:2", os.urandom(16))], transactional=True, producer_id=3),
        kw.control_batch(3, 3, kw.COMMIT),
    ])
    outcomes = {"ok": 0, "error": 0}
    for trial in range(400):
        data = bytearray(good)
        for _ in range(rng.randrange(1, 6)):
            if len(data) < 2:
                break
            how = rng.random()
            if how < 0.6:
                data[rng.randrange(len(data))] ^= 1 << rng.randrange(8)
            elif how < 0.8:
                del data[rng.randrange(len(data)):]
            else:
                pos = rng.randrange(len(data))
                data[pos:pos] = os.urandom(rng.randrange(1, 9))
        with EventsTopicIngest() as g:
            try:
                g.feed(bytes(data))
                g.drain_records()
                outcomes["ok"] += 1
            except IngestError:
                outcomes["error"] += 1
    assert outcomes["error"] > 100  # CRC-32C catches corruption; truncation just waits for more bytes


# ---- state-topic writer <-> events-topic reader, and the LZ4 pair, over random inputs -----------------------------------
_blob = st.one_of(st.binary(max_size=40), st.binary(min_size=200, max_size=600),
                  st.builds(lambda c, n: bytes([c]) * n, st.integers(0, 255), st.integers(0, 3000)))


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.text(alphabet="abcdefghij-:0123456789é✓", max_size=18), st.one_of(st.none(), _blob), st.integers(0, 2), st.booleans()),
                min_size=1, max_size=120),
       st.sampled_from(["none", "lz4"]), st.integers(1, 50), st.integers(64, 4000))
def test_whatever_the_writer_frames_the_reader_delivers_in_order(rows, codec, max_records, max_bytes):
    """rows: (key, value | None = tombstone, partition, skip).  Batches close on a record or a byte limit, with or
    without LZ4; every partition's log must decode to exactly the published rows, offsets consecutive."""
    from surge_amd.snapshot import RecordBatchWriter

    keys = [k.encode("utf-8") for k, *_ in rows]
    vals = [b"" if v is None else v for _, v, *_ in rows]
    kind = np.array([0 if skip else (2 if v is None else 1) for _, v, _, skip in rows], dtype=np.uint8)
    part = np.array([p for *_, p, _ in rows], dtype=np.int32)
    key_off = np.cumsum([0] + [len(k) for k in keys])
    val_off = np.cumsum([0] + [len(v) for v in vals])
    kbuf = np.frombuffer(b"".join(keys) or b"\0", np.uint8)
    vbuf = np.frombuffer(b"".join(vals) or b"\0", np.uint8)
    with RecordBatchWriter(3, max_records_per_batch=max_records, max_batch_bytes=max_bytes, compression=codec) as w:
        w.append(kind, part, kbuf, key_off, vbuf, val_off, timestamp_ms=1)
        w.append(kind, part, kbuf, key_off, vbuf, val_off, timestamp_ms=2)  # the logs continue across appends
        logs = [w.partition_bytes(p) for p in range(3)]
    for p, (wire, n_rec, next_off) in enumerate(logs):
        want = [(keys[i], None if kind[i] == 2 else vals[i]) for _ in range(2) for i in range(len(rows)) if kind[i] != 0 and part[i] == p]
        assert n_rec == len(want) == next_off
        with EventsTopicIngest() as g:
            g.feed(wire)
            got = g.drain_records()
        # a record with an empty key and an empty value is the reference producer's "flush" marker: the reader drops it
        want_delivered = [(i, k, v) for i, (k, v) in enumerate(want) if not (k == b"" and v == b"")]
        assert [(o, k if k is not None else b"", v) for o, _, k, v in got] == want_delivered


@settings(max_examples=80, deadline=None)
@given(st.lists(_blob, max_size=12).map(b"".join))
def test_lz4_pair_agrees_with_liblz4_in_both_directions(data):
    import ctypes

    import pytest as _pytest

    pa = _pytest.importorskip("pyarrow")
    from surge_amd import _native

    L = _native.load()
    cap = L.surge_lz4_frame_bound(len(data))
    dst = ctypes.create_string_buffer(cap)
    n = L.surge_lz4_frame_compress(data, len(data), dst, cap)
    assert 0 < n <= cap
    if data:
        assert pa.decompress(dst.raw[:n], decompressed_size=len(data), codec="lz4", asbytes=True) == data
    theirs = pa.compress(data, codec="lz4", asbytes=True)
    out = ctypes.create_string_buffer(len(data) + 1)
    assert L.surge_lz4_frame_decompress(theirs, len(theirs), out, len(data) + 1) == len(data) and out.raw[: len(data)] == data
