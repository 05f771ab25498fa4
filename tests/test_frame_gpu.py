"""-m gpu: the device framer (surge_device_framer_*, surge_amd/csrc/frame_kernels.hip) against the host record-batch writer
(surge_snapshot_writer_*, itself checked against the independent test-side writer in tests/test_ingest.py): the same input
must give the same bytes — records, batch cuts, headers, CRCs, offsets across publishes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_input(rng, n, n_part, key_max, val_max, p_skip=0.5, p_tomb=0.1, key_min=0):
    p_tomb = min(p_tomb, 1 - p_skip)
    kind = rng.choice([0, 1, 2], size=n, p=[p_skip, max(0.0, 1 - p_skip - p_tomb), p_tomb]).astype(np.uint8)
    part = rng.integers(0, n_part, size=n).astype(np.int32)
    klen = rng.integers(key_min, key_max + 1, size=n)
    vlen = np.where(kind == 1, rng.integers(0, val_max + 1, size=n), 0)  # the filtered encoder writes text for VALUE aggregates only
    key_off = np.zeros(n + 1, np.int64); np.cumsum(klen, out=key_off[1:])
    val_off = np.zeros(n + 1, np.int64); np.cumsum(vlen, out=val_off[1:])
    keys = rng.integers(32, 127, size=max(int(key_off[-1]), 1)).astype(np.uint8)
    vals = rng.integers(32, 127, size=max(int(val_off[-1]), 1)).astype(np.uint8)
    return kind, part, keys, key_off, vals, val_off


def host_frames(writer, inp, ts):
    kind, part, keys, key_off, vals, val_off = inp
    writer.reset()
    writer.append(kind, part, keys, key_off, vals, val_off, ts)
    out = {}
    for p in range(writer.n_partitions):
        data, nrec, _ = writer.partition_bytes(p)
        if nrec:
            out[p] = data
    return out


def device_frames(framer, inp, ts):
    import torch

    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a).to(dev) for a in inp]
    torch.cuda.synchronize(dev)
    return {p: bytes(v) for p, v in framer.frame(*t, timestamp_ms=ts).items()}


@pytest.mark.parametrize("n,n_part,max_records,max_bytes,key_max,val_max", [
    (0, 3, 0, 0, 8, 40), (1, 1, 0, 0, 8, 40), (500, 4, 0, 0, 12, 120), (5000, 7, 1, 0, 5, 30), (5000, 3, 3, 0, 5, 30),
    (6000, 2, 70, 0, 20, 60), (6000, 5, 0, 300, 10, 90), (6000, 1, 0, 5000, 3, 50), (40000, 2, 0, 0, 13, 100),
    (30000, 1, 0, 0, 2, 61),      # bodies around 63 / 64 bytes: the length prefix goes from one byte to two
    (30000, 1, 20000, 1 << 30, 1, 3),  # 15 k tiny records in one batch: offsetDelta crosses 64 and 8192
    (200000, 64, 0, 0, 13, 100),
])
def test_device_framer_writes_the_host_writers_bytes(n, n_part, max_records, max_bytes, key_max, val_max):
    from surge_amd.snapshot import DeviceFramer, RecordBatchWriter

    rng = np.random.default_rng(n + 31 * n_part + max_records + max_bytes)
    with RecordBatchWriter(n_part, max_records, max_bytes) as w, DeviceFramer(n_part, 0, max_records, max_bytes) as f:
        for publish in range(3):  # every partition's log continues from publish to publish
            inp = make_input(rng, n, n_part, key_max, val_max, p_skip=[0.5, 0.9, 0.0][publish])
            exp = host_frames(w, inp, 1_700_000_000_000 + publish)
            got = device_frames(f, inp, 1_700_000_000_000 + publish)
            assert sorted(got) == sorted(exp)
            for p in exp:
                assert got[p] == exp[p], (publish, p, len(got[p]), len(exp[p]))
            assert f.records == int(np.count_nonzero(inp[0]))
            nxt = [w.partition_bytes(p)[2] for p in range(n_part)]
            assert list(f.next_offsets()) == nxt


def test_device_framer_output_is_read_back_by_the_ingest_and_rejects_bad_input():
    import torch

    from surge_amd.ingest import EventsTopicIngest
    from surge_amd.snapshot import DeviceFramer

    rng = np.random.default_rng(3)
    n, n_part = 3000, 5
    inp = make_input(rng, n, n_part, 9, 70, p_skip=0.3, key_min=1)  # (the reader treats a record without a key as a flush record)
    kind, part, keys, key_off, vals, val_off = inp
    with DeviceFramer(n_part) as f:
        got = device_frames(f, inp, 123)
        seen = 0
        for p, data in got.items():
            with EventsTopicIngest() as g:  # CRC, framing and varints are checked by the reader
                g.feed(data)
                recs = g.drain_records()
            idx = [a for a in range(n) if kind[a] and part[a] == p]
            assert len(recs) == len(idx)
            for (offset, _, k, v), a in zip(recs, idx):
                assert k == keys[key_off[a]:key_off[a + 1]].tobytes()
                assert v == (vals[val_off[a]:val_off[a + 1]].tobytes() if kind[a] == 1 else None)
            assert [r[0] for r in recs] == list(range(len(idx)))
            seen += len(recs)
        assert seen == int(np.count_nonzero(kind))
        before = f.next_offsets().copy()
        bad = list(inp)
        bad[1] = part.copy(); bad[1][np.nonzero(kind)[0][7]] = n_part  # a partition that does not exist
        with pytest.raises(RuntimeError):
            device_frames(f, tuple(bad), 124)
        assert list(f.next_offsets()) == list(before)  # nothing advanced
        bad[1] = part
        bad[0] = kind.copy(); bad[0][5] = 9  # an unknown kind
        with pytest.raises(RuntimeError):
            device_frames(f, tuple(bad), 125)
        assert device_frames(f, inp, 126) != {}
