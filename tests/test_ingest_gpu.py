"""N1 on the device (include/surge_ingest.h, "device decode"): the host frames the record batches (headers, CRC-32C,
read_committed, LZ4), the GPU parses the records, interns the aggregate ids and decodes the event values.  Held to the
HOST decoder of the same library on the same wire bytes (which the CPU suite holds to the independent test-side writer,
liblz4, the protobuf runtime's varints ...): same records in the same order, same aggregate numbering, same key table,
same events — and, through the fold, to the oracle."""
import json
import os
import random
import struct
import uuid

import numpy as np
import pytest

import kafka_wire as kw
from fixture_models import (BA_CREATED, BA_UPDATED, BankAccountCommandModel, BankAccountCreated, BankAccountUpdated, CounterBusinessLogic,
                            CountDecremented, CountIncremented, NoOpEvent)
from oracle import oracle
from surge_amd import schema as S
from surge_amd.ingest import READ_COMMITTED, READ_UNCOMMITTED, DeviceDecoder, EventJsonTemplate, EventsTopicIngest, IngestError


def counter_event(ty, seq, arg):
    e = np.zeros(1, dtype=S.EVENT_DTYPE)
    e["type"], e["seq"], e["raw"] = ty, seq, np.uint64(np.uint32(np.int32(arg)))
    return e.tobytes()


# ---- CPU: the framing mode of the host decoder ----------------------------------------------------------------------
def test_frames_mode_hands_out_the_records_sections_of_deliverable_batches_only():
    ev = lambda seq: counter_event(S.EVT_INC, seq, 1)
    b0 = kw.record_batch(0, [(b"a:1", ev(1))], transactional=True, producer_id=7)
    b1 = kw.record_batch(1, [(b"b:1", ev(1)), (b"b:2", ev(2))], transactional=True, producer_id=9, compression="lz4")
    b2 = kw.record_batch(3, [(b"c:1", ev(1)), (b"", b"")])
    with EventsTopicIngest(READ_COMMITTED, frames=True) as g:
        g.feed(b0 + b1 + b2 + kw.control_batch(5, 9, kw.ABORT))
        assert g.ready == 0  # producer 7's transaction is still open: nothing behind it is stable
        with pytest.raises(IngestError):
            g.drain_fixed16()  # a framing decoder does not parse records
        g.feed(kw.control_batch(6, 7, kw.COMMIT))
        sections, arena = g.drain_sections()
        assert [(int(s["base_offset"]), int(s["n_records"])) for s in sections] == [(0, 1), (3, 2)]  # the aborted batch is gone
        import ctypes

        # a section is the batch's records section, verbatim (uncompressed)
        want = kw.record(0, b"a:1", ev(1))
        assert ctypes.string_at(arena + int(sections[0]["byte_off"]), int(sections[0]["byte_len"])) == want
        c = g.counters()
        assert c["records_aborted"] == 2 and c["control_batches"] == 2 and c["records_delivered"] == 3
    with EventsTopicIngest(READ_COMMITTED) as g, pytest.raises(IngestError):
        g.drain_sections()


def test_the_host_f64_parser_is_correctly_rounded():
    """surge_parse_f64_json = Eisel-Lemire (the code the device runs) + strtod for what it cannot decide: against
    Python's float(), which is correctly rounded — shortest-round-trip spellings, arbitrary digit strings, ties, the
    subnormal / overflow edges."""
    import ctypes

    from surge_amd import _native

    lib = _native.load()
    rng = np.random.default_rng(3)
    rnd = random.Random(3)

    def parse(txt):
        b = ctypes.c_uint64()
        rc = lib.surge_parse_f64_json(txt.encode(), len(txt), ctypes.byref(b))
        return rc, b.value

    texts = [repr(float(x)) for x in rng.integers(0, 0x7FF0000000000000, size=30000, dtype=np.uint64).view(np.float64)]
    texts += [repr(float(x)) for x in rng.random(30000) * 1e6]
    for _ in range(30000):
        nd = rnd.randint(1, 19)
        ds = "".join(rnd.choice("0123456789") for _ in range(nd))
        texts.append(f"{ds}e{rnd.randint(-345, 310)}")
        pos = rnd.randint(0, nd)
        texts.append((ds[:pos] or "0") + "." + (ds[pos:] or "0"))
    for e in range(-345, 312):
        texts += [f"1e{e}", f"9.999999999999999e{e}", f"1.0000000000000002E{e}"]
    texts += ["0", "-0", "0.0", "2.2250738585072011e-308", "2.2250738585072014e-308", "4.9e-324", "2.4703282292062327e-324",
              "2.4703282292062328e-324", "1.7976931348623157e308", "1.7976931348623159e308", "9007199254740993", "9007199254740992.5",
              "1e400", "-1e-400"]
    texts += [f"{rnd.randint(2 ** 52, 2 ** 53 - 1)}.5" for _ in range(2000)]  # exactly between two doubles: ties to even
    slow = 0
    for t in texts:
        rc, b = parse(t)
        assert rc in (0, 1) and b == int(np.float64(float(t)).view(np.uint64)), (t, rc)
        slow += rc
    assert slow == 0  # all of these have at most 19 digits: the fast path decided every one
    rc, b = parse("0.1000000000000000055511151231257827021181583404541015625")  # 55 digits: decided by strtod
    assert rc == 1 and b == int(np.float64(0.1).view(np.uint64))
    for t in ["", "abc", "1.", ".5", "1e", "--1", "1.5.2", "0x10"]:
        assert parse(t)[0] == -7


# ---- GPU ------------------------------------------------------------------------------------------------------------
def both_decoders(wire, template=None, isolation=READ_COMMITTED, chunks=None, device_lz4=False):
    """The same wire bytes through the host decoder and through host framing + the device decoder (device_lz4: LZ4 frames
    are left for the GPU too)."""
    with EventsTopicIngest(isolation) as g:
        g.feed(wire)
        host = g.drain_json(template) if template is not None else g.drain_fixed16()
        host_keys = g.key_table().keys
    with EventsTopicIngest(isolation, frames=True, device_lz4=device_lz4) as g, DeviceDecoder(template) as d:
        if chunks is None:
            g.feed(wire)
            d.push_from(g)
        else:  # several feeds / pushes: the key table and the result grow across pushes
            pos = 0
            for c in chunks:
                g.feed(wire[pos:pos + c])
                pos += c
                d.push_from(g)
            g.feed(wire[pos:])
            d.push_from(g)
        agg, ev, off, n_keys = d.result()
        dev = (agg.cpu().numpy(), ev.cpu().numpy().view(S.EVENT_DTYPE).reshape(-1), off.cpu().numpy())
        dev_keys = d.keys()
        counters = d.counters()
    assert n_keys == len(dev_keys)
    return host, host_keys, dev, dev_keys, counters


@pytest.mark.gpu
def test_device_decoder_equals_the_host_decoder_on_fixed16_topics_with_transactions_lz4_headers_and_flush_records():
    rng = random.Random(11)
    batches, off = [], 0
    pid = 100
    for b in range(120):
        n = rng.randrange(1, 40)
        rs = []
        for j in range(n):
            if rng.random() < 0.03:
                rs.append((b"", b"", []))  # the producer's flush record
                continue
            agg = f"acct-{rng.randrange(300):05d}" if rng.random() < 0.9 else "ünï-✓-" + "x" * rng.randrange(0, 40)
            hdrs = [(b"aggregate_id", agg.encode())] if rng.random() < 0.3 else []
            rs.append((f"{agg}:{rng.randrange(10 ** 6)}".encode(), counter_event(rng.choice([0, 1, 2]), off + j, rng.randrange(-50, 50)), hdrs))
        txn = rng.random() < 0.5
        batches.append(kw.record_batch(off, rs, compression=rng.choice(["none", "lz4"]), transactional=txn, producer_id=pid if txn else -1))
        off += n
        if txn:
            batches.append(kw.control_batch(off, pid, kw.COMMIT if rng.random() < 0.8 else kw.ABORT))
            off += 1
            pid += 1
    wire = b"".join(batches)
    for chunks, device_lz4 in ((None, False), ([len(wire) // 3, len(wire) // 3], False), (None, True), ([len(wire) // 2], True)):
        host, host_keys, dev, dev_keys, counters = both_decoders(wire, chunks=chunks, device_lz4=device_lz4)
        assert dev_keys == host_keys
        for h, g in zip(host, dev):
            assert h.shape == g.shape and h.tobytes() == g.tobytes()
        assert counters["records_delivered"] == host[0].shape[0] and counters["flush_records_skipped"] > 0


@pytest.mark.gpu
def test_device_decoder_key_interning_across_table_growth_prefix_ids_and_many_pushes():
    rng = random.Random(5)
    ids = [f"agg-{i}" for i in range(20000)] + ["a" * k for k in range(1, 801)] + [""]
    seq = ids * 3
    rng.shuffle(seq)
    wire, off = [], 0
    for s0 in range(0, len(seq), 700):
        chunk = seq[s0:s0 + 700]
        wire.append(kw.record_batch(off, [(f"{a}:{j}".encode(), counter_event(1, j, 1)) for j, a in enumerate(chunk)]))
        off += len(chunk)
    wire = b"".join(wire)
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, chunks=[len(wire) // 7] * 6)
    assert dev_keys == host_keys == list(dict.fromkeys(seq))  # first-delivered order, every id once, prefixes distinct
    assert host[0].tobytes() == dev[0].tobytes() and host[2].tobytes() == dev[2].tobytes()


@pytest.mark.gpu
def test_device_decoder_decodes_play_json_counter_and_bank_account_events_like_the_host_decoder():
    rng = random.Random(7)
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    recs = []
    for i in range(6000):
        agg = f"agg-{rng.randrange(400)}"
        e = rng.choice([CountIncremented(agg, rng.randrange(-2 ** 31, 2 ** 31), i + 1), CountDecremented(agg, rng.randrange(0, 1000), i + 1), NoOpEvent(agg, i + 1)])
        m = fmt.write_event(e)
        recs.append((m.key.encode(), m.value))
    wire = b"".join(kw.record_batch(s, recs[s:s + 500], compression="lz4" if s % 1000 else "none") for s in range(0, len(recs), 500))
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, model.event_json_template())
    assert dev_keys == host_keys
    for h, g in zip(host, dev):
        assert h.tobytes() == g.tobytes()
    # the same records already framed (a JVM consumer's key / value arrays): surge_device_decoder_push_records, in two
    # polls, then folded onto a store with surge_replay_append_decoded
    import ctypes

    from surge_amd.replay import ReplayEngine

    with DeviceDecoder(model.event_json_template()) as d, ReplayEngine(model.event_algebra()) as eng:
        eng.load_csr(np.zeros(1, np.int64), np.zeros(0, dtype=S.EVENT_DTYPE))
        eng.fold()
        half = len(recs) // 2
        d.push_records([k for k, _ in recs[:half]] + [b""], [v for _, v in recs[:half]] + [b""], list(range(half + 1)))  # + a flush record
        d.push_records([k for k, _ in recs[half:]], [v for _, v in recs[half:]], list(range(half, len(recs))))
        agg, ev, off, n_keys = d.result()
        assert d.keys() == host_keys and agg.cpu().numpy().tobytes() == host[0].tobytes()
        assert ev.cpu().numpy().view(S.EVENT_DTYPE).reshape(-1).tobytes() == host[1].tobytes() and off.cpu().numpy().tobytes() == host[2].tobytes()
        assert d.counters()["flush_records_skipped"] == 1
        n_ev, n_k = ctypes.c_int64(), ctypes.c_int64()
        assert d._lib.surge_replay_append_decoded(eng._h, d._h, ctypes.byref(n_ev), ctypes.byref(n_k)) == 0
        assert (n_ev.value, n_k.value) == (len(recs), len(host_keys)) and d.result()[0].shape[0] == 0
        eng.n_agg = n_k.value
        order = np.argsort(host[0], kind="stable")
        so = np.zeros(n_k.value + 1, np.int64)
        np.cumsum(np.bincount(host[0], minlength=n_k.value), out=so[1:])
        assert eng.snapshot().tobytes() == oracle.fold_csr(so, host[1][order], None, model.event_algebra()).tobytes()

    # BankAccount: Doubles as play-json writes them (and a few spellings it never writes), field order shuffled, extra
    # fields, nested values to skip
    ba = BankAccountCommandModel()
    tmpl = ba.event_json_template()
    from surge_amd.encode import play_json_double

    nprng = np.random.default_rng(9)
    recs = []
    vals = np.concatenate([np.round(nprng.random(3000) * 1e7) / 100, nprng.random(2000) * 10.0 ** nprng.integers(-12, 25, size=2000),
                           nprng.integers(0, 0x7FF0000000000000, size=2000, dtype=np.uint64).view(np.float64),
                           np.array([0.0, 5e-324, 1.7976931348623157e308, 0.1 + 0.2, 1e21, 100.0, 1e-7])])
    for i, v in enumerate(vals):
        acct = str(uuid.UUID(int=rng.randrange(1 << 100) % 500 + 1))
        text = play_json_double(float(v))
        if i % 3 == 0:
            fields = [('"accountNumber"', f'"{acct}"'), ('"accountOwner"', '"Jane \\"J\\" Doe"'), ('"securityCode"', '"1234"'), ('"balance"', text),
                      ('"_type"', '"docs.command.BankAccountCreated"')]
        else:
            fields = [('"_type"', '"docs.command.BankAccountUpdated"'), ('"accountNumber"', f'"{acct}"'), ('"newBalance"', text),
                      ('"audit"', '{"by":["x",{"y":"}"}],"n":null,"ok":true}')]
        if i % 5 == 0:
            rng.shuffle(fields)
        value = ("{" + ",".join(f"{k}:{v2}" for k, v2 in fields) + "}").encode()
        recs.append((f"{acct}:{i}".encode(), value))
    recs.append((b"long:1", b'{"_type":"docs.command.BankAccountUpdated","newBalance":0.1000000000000000055511151231257827021181583404541015625}'))
    recs.append((b"long:2", b'{"_type":"docs.command.BankAccountUpdated","newBalance":-12345678901234567890123.5E-3}'))
    wire = b"".join(kw.record_batch(s, recs[s:s + 400]) for s in range(0, len(recs), 400))
    host, host_keys, dev, dev_keys, counters = both_decoders(wire, tmpl)
    assert dev_keys == host_keys and counters["doubles_parsed_on_host"] == 2  # the two > 19-digit spellings
    for h, g in zip(host, dev):
        assert h.tobytes() == g.tobytes()
    got = dev[1]["raw"][: len(vals)].view(np.float64)
    assert got.tobytes() == vals.astype(np.float64).tobytes()  # text -> double is the inverse of double -> text, bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("per_batch", [1, 64, 65, 128, 129, 192, 193, 600])
def test_every_workgroup_size_of_the_record_kernel_decodes_like_the_host_decoder(per_batch):
    """section_kernel runs with 64 / 128 / 192 / 256 lanes — the smallest that takes the push's largest batch in one round
    (ingest_kernels.hip) — out of three LDS classes (sections up to 8 KiB, up to 16.25 KiB, the rest): batches of every size
    around the boundaries, lz4 and plain, small and long values, one push and several."""
    rng = random.Random(1000 + per_batch)
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    recs = []
    for i in range(max(3 * per_batch, 400)):
        agg = f"agg-{rng.randrange(90)}" + ("-" + "w" * rng.randrange(0, 150) if per_batch % 2 else "")  # (odd sizes: long ids — sections beyond 16 KiB)
        e = rng.choice([CountIncremented(agg, rng.randrange(-2 ** 31, 2 ** 31), i + 1), CountDecremented(agg, rng.randrange(0, 1000), i + 1), NoOpEvent(agg, i + 1)])
        m = fmt.write_event(e)
        recs.append((m.key.encode(), m.value))
    batches = [kw.record_batch(s, recs[s:s + per_batch], compression="lz4" if (s // per_batch) % 3 else "none") for s in range(0, len(recs), per_batch)]
    wire = b"".join(batches)
    for chunks, device_lz4 in ((None, True), ([len(wire) // 2], False)):
        host, host_keys, dev, dev_keys, _ = both_decoders(wire, model.event_json_template(), chunks=chunks, device_lz4=device_lz4)
        assert dev_keys == host_keys
        for h, g in zip(host, dev):
            assert h.shape == g.shape and h.tobytes() == g.tobytes()


@pytest.mark.gpu
def test_the_fast_json_walk_accepts_and_decodes_exactly_what_the_host_decoder_does_on_mutated_values():
    """The device decodes event values with a table-driven automaton (one loop over the bytes, in lockstep) that only ever
    accepts, and falls back to the exact walk otherwise.  Held to the host decoder on values mutated every which way:
    whitespace, reordered / duplicated / escaped field names, nested values, odd number spellings, truncations, stray
    bytes — every value the host accepts decodes to the same event, every value it rejects fails the push."""
    rng = random.Random(13)
    tmpl = CounterBusinessLogic().command_model().event_json_template()
    from surge_amd.ingest import IngestError as IE

    def base():
        agg, seq, arg = f"a{rng.randrange(50)}", rng.randrange(-5, 2 ** 31 + 3), rng.randrange(-2 ** 31 - 3, 2 ** 31 + 3)  # (a few beyond Int)
        kind = rng.randrange(3)
        fields = [("aggregateId", json.dumps(agg))]
        if kind == 0:
            fields.append(("incrementBy", str(arg)))
        if kind == 1:
            fields.append(("decrementBy", str(arg)))
        fields += [("sequenceNumber", str(seq)), ("_type", json.dumps(["countIncremented", "countDecremented", "no-op"][kind]))]
        return fields

    def render(fields):
        ws = lambda: rng.choice(["", "", "", " ", "\n", "\t "])  # noqa: E731
        body = ",".join(f'{ws()}"{k}"{ws()}:{ws()}{v}{ws()}' for k, v in fields)
        return (ws() + "{" + body + "}" + ws()).encode()

    values = []
    for _ in range(6000):
        f = base()
        r = rng.random()
        if r < 0.15:
            rng.shuffle(f)
        elif r < 0.25:
            f.insert(rng.randrange(len(f) + 1), rng.choice([("extra", '{"a":[1,"}",{"b":null}],"c":"\\\""}'), ("sequenceNumber", '"7"'), ("sequenceNumber", "9"),
                                                             ("_type", '"no-op"'), ("_type", "5"), ("incrementBy", "1e3"), ("incrementBy", "+5"), ("x", "true"),
                                                             ("se\\u0071uenceNumber", "4"), ("y", "[]"), ("z", ".5"), ("w", "-"), ("v", "tru"), ("u", "nul1")]))
        elif r < 0.30:
            f = f * 7  # more than 24 fields
        v = render(f)
        r = rng.random()
        if r < 0.12:  # a byte-level mutation
            b = bytearray(v)
            k = rng.randrange(len(b))
            op = rng.randrange(3)
            if op == 0:
                del b[k]
            elif op == 1:
                b.insert(k, rng.choice(b' {}[]",:\\\\0-e.tx'))
            else:
                b[k] = rng.choice(b' {}[]",:\\\\0-e.tx')
            v = bytes(b)
        elif r < 0.15:
            v = v[: rng.randrange(len(v))]
        values.append(v)
    good, bad, want = [], [], []
    for v in values:
        try:
            want.append(tmpl.decode(v))
            good.append(v)
        except IE:
            bad.append(v)
    assert len(good) > 4000 and len(bad) > 300
    with DeviceDecoder(tmpl) as d:
        d.push_records([b"k%d:1" % (i % 97) for i in range(len(good))], good)
        ev = d.result()[1].cpu().numpy().view(S.EVENT_DTYPE).reshape(-1)
        assert ev.tobytes() == np.array(want, dtype=S.EVENT_DTYPE).tobytes()
        for v in bad[:120]:
            with pytest.raises(IE):
                d.push_records([b"k:1", b"j:1"], [good[0], v])
        assert d.result()[0].shape[0] == len(good)  # the failed pushes appended nothing
    # the same through the wire path (values staged in LDS)
    wire = b"".join(kw.record_batch(s0, [(b"k%d:1" % (i % 97), v) for i, v in enumerate(good[s0:s0 + 150], s0)]) for s0 in range(0, len(good), 150))
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, tmpl)
    assert dev_keys == host_keys and all(h.tobytes() == g.tobytes() for h, g in zip(host, dev)) and dev[1].tobytes() == np.array(want, dtype=S.EVENT_DTYPE).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("bad,why", [
    (b'{"_type":"docs.command.Nope","newBalance":1}', "event type"),
    (b'{"_type":"docs.command.BankAccountUpdated"}', "field"),
    (b'{"_type":"docs.command.BankAccountUpdated","newBalance":"12"}', "field"),
    (b'{"_type":"docs.command.BankAccountUpdated","newBalance":1', "JSON"),
    (b'[1,2]', "JSON"),
])
def test_device_decoder_rejects_what_the_host_decoder_rejects_and_names_the_offset(bad, why):
    tmpl = BankAccountCommandModel().event_json_template()
    good = b'{"_type":"docs.command.BankAccountUpdated","newBalance":1.5}'
    wire = kw.record_batch(40, [(b"a:1", good), (b"b:1", bad), (b"c:1", good)])
    with EventsTopicIngest() as g:
        g.feed(wire)
        with pytest.raises(IngestError):
            g.drain_json(tmpl)
    with EventsTopicIngest(frames=True) as g, DeviceDecoder(tmpl) as d:
        g.feed(wire)
        with pytest.raises(IngestError) as ei:
            d.push_from(g)
        assert ei.value.status == -7 and "offset 41" in str(ei.value) and why in str(ei.value)
        assert d.result()[0].shape[0] == 0  # nothing of the failing push was appended
    # fixed-16 topics: wrong value size, null value, a record whose length runs past the batch
    with EventsTopicIngest(frames=True) as g, DeviceDecoder() as d:
        g.feed(kw.record_batch(0, [(b"k:1", b"not sixteen bytes")]))
        with pytest.raises(IngestError, match="16-byte"):
            d.push_from(g)
        g.feed(kw.record_batch(1, [(b"k:1", None)]))
        with pytest.raises(IngestError, match="null"):
            d.push_from(g)
        sec = np.zeros(1, dtype=[("byte_off", "<i8"), ("byte_len", "<i8"), ("base_offset", "<i8"), ("n_records", "<i4"), ("codec", "<i4")])
        body = np.frombuffer(kw.record(0, b"k:1", counter_event(1, 1, 1))[:-3], dtype=np.uint8).copy()  # truncated record
        sec["byte_len"], sec["n_records"] = body.shape[0], 1
        with pytest.raises(IngestError, match="malformed"):
            d.push(sec, body.ctypes.data)


@pytest.mark.gpu
def test_a_failed_push_leaves_the_decoder_exactly_as_it_was():
    """a push that fails after its keys were probed must not leave them half-inserted (the next push would read
    an unassigned id) nor interned (ids would drift from the host decoder's first-delivered numbering).  The push is
    rolled back on the device: the same decoder then numbers a good topic exactly like a fresh one."""
    good1 = kw.record_batch(0, [(f"k{j}:1".encode(), counter_event(1, j, 1)) for j in range(50)])
    bad = kw.record_batch(50, [(b"new-a:1", counter_event(1, 1, 1)), (b"new-b:1", b"not sixteen bytes"), (b"k3:2", counter_event(1, 2, 1))])
    good2 = kw.record_batch(53, [(b"new-b:1", counter_event(2, 5, 7)), (b"k3:2", counter_event(1, 2, 1)), (b"new-a:1", counter_event(1, 1, 1))])
    with EventsTopicIngest(frames=True) as g, DeviceDecoder() as d:
        g.feed(good1)
        d.push_from(g)
        g.feed(bad)
        with pytest.raises(IngestError, match="16-byte"):
            d.push_from(g)
        assert d.keys() == [f"k{j}" for j in range(50)] and d.result()[0].shape[0] == 50  # nothing interned, nothing appended
        g.feed(good2)
        d.push_from(g)
        assert d.keys() == [f"k{j}" for j in range(50)] + ["new-b", "new-a"]  # first-delivered order of the push that went through
        agg = d.result()[0].cpu().numpy()
        assert agg[50:].tolist() == [50, 3, 51]
        assert d.stats()["pushes"] == 2 and d.stats()["hash_reseeds"] == 0


@pytest.mark.gpu
def test_a_key_hash_collision_is_handled_by_reseeding_the_table(monkeypatch):
    """two aggregate ids with the same 64-bit hash used to fail the push (detected, not handled).  With
    SURGE_INGEST_DEBUG_WEAK_HASH=1 the table's first hash function keeps 8 bits, so 2000 ids are certain to collide: the
    decoder detects it before anything is committed, gives the table another hash function (known keys re-hashed from
    the key arena, the push's records from their bytes) and runs the push again — same records, ids and key table as the
    host decoder, over several pushes, with keys from before and after the re-seed."""
    monkeypatch.setenv("SURGE_INGEST_DEBUG_WEAK_HASH", "1")
    rng = random.Random(3)
    ids = [f"agg-{i}" for i in range(2000)]
    seq = ids * 2
    rng.shuffle(seq)
    seq = ids[:3] + seq  # the first push (one batch) holds no collision yet: the re-seed happens with keys already interned
    wire, off = [kw.record_batch(0, [(f"{a}:{j}".encode(), counter_event(1, j, 1)) for j, a in enumerate(seq[:3])])], 3
    for s0 in range(3, len(seq), 500):
        chunk = seq[s0:s0 + 500]
        wire.append(kw.record_batch(off, [(f"{a}:{j}".encode(), counter_event(1, j, 1)) for j, a in enumerate(chunk)]))
        off += len(chunk)
    first = len(wire[0])
    wire = b"".join(wire)
    with EventsTopicIngest() as g:
        g.feed(wire)
        host = g.drain_fixed16()
        host_keys = list(g.key_table().keys)
    with EventsTopicIngest(frames=True) as g, DeviceDecoder() as d:
        g.feed(wire[:first])
        d.push_from(g)
        assert d.stats()["hash_reseeds"] == 0
        rest = wire[first:]
        for c0 in range(0, len(rest), len(rest) // 3 + 1):
            g.feed(rest[c0:c0 + len(rest) // 3 + 1])
            d.push_from(g)
        st = d.stats()
        assert st["hash_reseeds"] == 1 and st["hash_function"] == 1  # one re-seed took the weak function out of use
        agg, ev, off_d, n_keys = d.result()
        assert d.keys() == host_keys and n_keys == len(host_keys)
        assert agg.cpu().numpy().tobytes() == host[0].tobytes() and ev.cpu().numpy().tobytes() == host[1].tobytes()
        assert off_d.cpu().numpy().tobytes() == host[2].tobytes()


@pytest.mark.gpu
def test_push_records_refuses_offsets_that_run_backwards_before_any_launch():
    import ctypes

    from surge_amd import _native

    lib = _native.load()
    with DeviceDecoder() as d:
        keys = np.frombuffer(b"a:1b:1c:1", dtype=np.uint8)
        vals = np.frombuffer(counter_event(1, 1, 1) * 3, dtype=np.uint8)
        ko = np.array([0, 1_000_000, 5, 9], dtype=np.int64)  # an intermediate offset far outside the buffer
        vo = np.array([0, 16, 32, 48], dtype=np.int64)
        rc = lib.surge_device_decoder_push_records(d._h, keys.ctypes.data_as(ctypes.c_void_p), ko.ctypes.data_as(ctypes.c_void_p),
                                                   vals.ctypes.data_as(ctypes.c_void_p), vo.ctypes.data_as(ctypes.c_void_p), None, 3)
        assert rc == -1 and b"must not decrease" in lib.surge_device_decoder_last_error(d._h)
        d.push_records([b"a:1", b"b:1"], [counter_event(1, 1, 1)] * 2)  # the decoder is still usable
        assert d.keys() == ["a", "b"]


@pytest.mark.gpu
def test_lz4_frames_decoded_on_the_device_block_by_block():
    """SURGE_INGEST_DEVICE_LZ4: the records section of an lz4 batch stays one LZ4 frame; the host reads the frame header and
    the block size words, one wave per block decodes it.  Frames of every shape the writers here can produce — many
    blocks (batches of several hundred KB), stored blocks, long overlapping matches of every period, highly and barely
    compressible values, liblz4-written frames with and without a content size — against the host decoder; frames the
    device does not take block by block (256 KiB blocks) fall back to the host inside the decoder; damaged frames fail
    the push."""
    import pyarrow as pa

    rng = random.Random(31)
    nprng = np.random.default_rng(31)

    def value(kind, n):
        if kind == "zeros":
            return bytes(n)
        if kind == "random":
            return nprng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        if kind == "period":
            p = rng.randrange(1, 40)
            return (bytes(rng.randrange(256) for _ in range(p)) * (n // p + 1))[:n]
        words = [b"aggregateId", b"sequenceNumber", b"incrementBy", b"countIncremented", b"acct-", b'":"', b'","', b"0123456789"]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words)
        return bytes(out[:n])

    def arrow_lz4(raw):  # liblz4 (the reference implementation of the format), as bundled by Apache Arrow
        return pa.Codec("lz4").compress(raw, asbytes=True)

    batches, off = [], 0
    for b in range(60):
        n = rng.randrange(1, 30)
        kind = rng.choice(["zeros", "random", "period", "text", "text"])
        big = rng.random() < 0.2  # values that make the batch span several 64 KiB blocks
        rs = [(f"agg-{rng.randrange(200)}:{off + j}".encode(), value(kind, rng.randrange(40_000, 90_000) if big else rng.randrange(1, 400))) for j in range(n)]
        compressor = arrow_lz4 if b % 3 == 0 else None
        batches.append(kw.record_batch(off, rs, compression="lz4", compressor=compressor))
        off += n
    wire = b"".join(batches)
    with EventsTopicIngest() as g:
        g.feed(wire)
        host = g.drain_records()
        host_keys = g.key_table().keys
    for device_lz4 in (True, False):
        with EventsTopicIngest(frames=True, device_lz4=device_lz4) as g:
            g.feed(wire)
            sections, arena = g.drain_sections()
            assert set(int(c) for c in sections["codec"]) == ({3} if device_lz4 else {0})
            # (the values here are not events: check the records through the decoder's own metadata — keys, offsets, and
            # the fixed-16 decoder's verdict on sizes — by decoding a topic whose values ARE events below; here: lengths)
            total = sum(int(s["n_records"]) for s in sections)
            assert total == len(host)
    # the same shapes with 16-byte event values padded by record headers of every size, so that records straddle blocks
    batches, off = [], 0
    for b in range(80):
        n = rng.randrange(1, 60)
        rs = []
        for j in range(n):
            hdr_kind = rng.choice(["zeros", "random", "period", "text"])
            hdrs = [(b"h", value(hdr_kind, rng.randrange(0, 9000)))] if rng.random() < 0.7 else []
            rs.append((f"acct-{rng.randrange(500):04d}:{off + j}".encode(), counter_event(rng.choice([0, 1, 2]), off + j, rng.randrange(-9, 9)), hdrs))
        batches.append(kw.record_batch(off, rs, compression="lz4", compressor=arrow_lz4 if b % 4 == 0 else None))
        off += n
    wire = b"".join(batches)
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, device_lz4=True)
    assert dev_keys == host_keys
    for h, g2 in zip(host, dev):
        assert h.shape == g2.shape and h.tobytes() == g2.tobytes()
    # a frame with 256 KiB blocks (BD code 5): not kafka-clients' shape — decompressed on the host inside the decoder
    recs = [(f"k{j}:1".encode(), counter_event(1, j, j), [(b"h", value("text", 70_000))]) for j in range(6)]

    def frame_256k(raw):
        import struct as st

        flg, bd = 0x60, 0x50
        out = bytearray(st.pack("<I", 0x184D2204)) + bytes([flg, bd, kw.header_checksum(bytes([flg, bd]))])
        for s0 in range(0, len(raw), 262144):
            chunk = raw[s0:s0 + 262144]
            comp = kw.lz4_block_compress(chunk)
            out += st.pack("<I", len(comp)) + comp
        return bytes(out + st.pack("<I", 0))

    wire2 = kw.record_batch(0, recs, compression="lz4", compressor=frame_256k)
    host, host_keys, dev, dev_keys, _ = both_decoders(wire2, device_lz4=True)
    assert dev_keys == host_keys and host[1].tobytes() == dev[1].tobytes()
    # damage inside a compressed block (the batch CRC is recomputed so that the framing accepts it): the push fails
    good = bytearray(kw.record_batch(0, [(b"a:1", counter_event(1, 1, 1), [(b"h", value("text", 5000))])], compression="lz4"))
    body_at = 61 + 7 + 4  # batch header, frame header, first block's size word
    for flip in range(6):
        bad = bytearray(good)
        pos = body_at + rng.randrange(0, len(bad) - body_at - 8)
        bad[pos] ^= 0xFF
        crc = kw.crc32c(bytes(bad[21:]))
        struct.pack_into(">I", bad, 17, crc)
        with EventsTopicIngest(frames=True, device_lz4=True) as g, DeviceDecoder() as d:
            g.feed(bytes(bad))
            try:
                d.push_from(g)
                agg, ev, off2, _ = d.result()  # a flip inside a literal run still decodes: then it must equal the host's view
                with EventsTopicIngest() as gh:
                    gh.feed(bytes(bad))
                    h_agg, h_ev, h_off = gh.drain_fixed16()
                assert ev.cpu().numpy().view(S.EVENT_DTYPE).reshape(-1).tobytes() == h_ev.tobytes()
            except IngestError as e:
                assert e.status == -7


@pytest.mark.gpu
def test_a_full_lz4_block_that_ends_in_an_empty_last_sequence_decodes_like_on_the_host():
    """a 64 KiB block whose last match runs to the block's end, followed by a token with no literals (encoders
    other than liblz4 write that): the empty sequence's position, 65536, does not fit a table entry's 16 bits — it gets no
    entry.  The device result equals the host decoder's."""
    import struct as st

    def section(hv_len):
        hv = (b"0123456789abcdefghijklmnopqrstuvwxyz" * 2000)[: hv_len - 30000] + b"ab" * 15000
        return kw.record_batch(0, [(b"acct-0001:1", counter_event(1, 1, 5), [(b"h", hv)])])[61:]

    hv_len = 65536 - 60
    while len(section(hv_len)) != 65536:
        hv_len += 65536 - len(section(hv_len))
    raw_expected = section(hv_len)
    assert raw_expected.endswith(b"ab" * 15000)

    def odd_encoder(raw):
        assert raw == raw_expected
        lit, ml = 65536 - 29000, 29000  # the last 29000 bytes repeat with period 2

        def ext(v):
            out = bytearray()
            while v >= 255:
                out.append(255)
                v -= 255
            out.append(v)
            return bytes(out)

        block = bytes([0xFF]) + ext(lit - 15) + raw[:lit] + st.pack("<H", 2) + ext(ml - 4 - 15) + b"\x00"
        flg, bd = 0x60, 0x40
        return st.pack("<I", 0x184D2204) + bytes([flg, bd, kw.header_checksum(bytes([flg, bd]))]) + st.pack("<I", len(block)) + block + st.pack("<I", 0)

    wire = kw.record_batch(0, [(b"acct-0001:1", counter_event(1, 1, 5), [(b"h", raw_expected[-hv_len:])])], compression="lz4", compressor=odd_encoder)
    wire += kw.record_batch(1, [(b"acct-0002:1", counter_event(1, 1, 7))], compression="lz4")
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, device_lz4=True)
    assert dev_keys == host_keys == ["acct-0001", "acct-0002"]
    for h, g2 in zip(host, dev):
        assert h.tobytes() == g2.tobytes()


@pytest.mark.gpu
def test_events_topic_bytes_to_states_without_the_host_touching_a_record():
    """Kafka bytes -> host framing -> device decode -> device group-by + fold (K3) -> states, against the oracle's fold of
    the published events; the GPU state encoder takes the decoder's DEVICE key table as it is."""
    import torch

    from surge_amd.encode import JsonTemplate, encode_states
    from surge_amd.replay import ReplayEngine

    rng = random.Random(21)
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    published, recs = {}, []
    seqs = {}
    for i in range(30000):
        agg = f"agg-{int(rng.paretovariate(1.2)) % 2000}"
        seqs[agg] = seqs.get(agg, 0) + 1
        e = rng.choice([CountIncremented(agg, rng.randrange(1, 9), seqs[agg]), CountDecremented(agg, rng.randrange(1, 9), seqs[agg]), NoOpEvent(agg, seqs[agg])])
        m = fmt.write_event(e)
        recs.append((m.key.encode(), m.value))
        published.setdefault(agg, []).append(e)
    wire = b"".join(kw.record_batch(s, recs[s:s + 600], compression="lz4") for s in range(0, len(recs), 600))
    with EventsTopicIngest(frames=True) as g, DeviceDecoder(model.event_json_template()) as d, ReplayEngine(model.event_algebra()) as eng:
        pos, n_agg = 0, 0
        eng.load_csr(np.zeros(1, np.int64), np.zeros(0, dtype=S.EVENT_DTYPE))  # an empty store to grow
        eng.fold()
        while pos < len(wire):  # fetch by fetch: decode on the device, fold onto the resident state
            g.feed(wire[pos:pos + 200_000])
            pos += 200_000
            if d.push_from(g) == 0:
                continue
            agg, ev, _, n_keys = d.result()
            if n_keys > n_agg:
                eng.grow(n_keys)
                n_agg = n_keys
            eng.append_events(agg, ev)
            eng.synchronize()
            d.clear()
        states = eng.snapshot()
        keys = d.keys()
        assert sorted(keys) == sorted(published)
        for a, key in enumerate(keys):
            evs = model.encode_events(published[key])
            off = np.array([0, evs.shape[0]], np.int64)
            assert states[a].tobytes() == oracle.fold_csr(off, evs, None, model.event_algebra())[0].tobytes(), key
        # the decoder's device key table feeds the GPU state encoder directly
        import ctypes

        pk, po = ctypes.c_void_p(), ctypes.c_void_p()
        assert d._lib.surge_device_decoder_key_table(d._h, ctypes.byref(pk), ctypes.byref(po)) == 0
        n_bytes = sum(len(k.encode()) for k in keys)

        def view(ptr, count, typ):
            iface = {"shape": (count,), "typestr": typ, "data": (ptr.value, False), "version": 2}
            return torch.as_tensor(type("_S", (), {"__cuda_array_interface__": iface})(), device="cuda")

        d_out, d_off = encode_states(eng, JsonTemplate.counter(), view(pk, n_bytes, "|u1"), view(po, len(keys) + 1, "<i8"))
        out, offs = d_out.cpu().numpy().tobytes(), d_off.cpu().numpy()
        for a in (0, len(keys) // 2, len(keys) - 1):
            o = json.loads(out[offs[a]:offs[a + 1]])
            assert o["aggregateId"] == keys[a] and o["count"] == int(states[a]["count"])


@pytest.mark.gpu
@pytest.mark.parametrize("weak_hash", [False, True])
def test_two_host_threads_drive_one_decoder_and_the_fold_takes_the_results_without_a_host_wait(monkeypatch, weak_hash):
    """The recovery pipeline's threading contract (include/surge_ingest.h): ONE thread enqueues stage 1 of up to four fetches
    ahead (``PushPipeline``'s worker: ``push_async``), ONE other finishes the oldest push without the closing wait
    (``finish(wait=False)``) and hands the results to the engine — on a stream of its own — with
    ``surge_replay_append_decoded_async`` (events order the two streams; slots and result arrays are reused behind events,
    not behind host waits).  40 fetches through 5 slots: same key table and same states as the same bytes pushed one at a
    time with a host wait after every step — also when the table is re-seeded (weak first hash function) while later
    fetches, hashed with the old function, are already in flight."""
    from surge_amd.ingest import FramedFetches, PushPipeline
    from surge_amd.replay import ReplayEngine

    if weak_hash:
        monkeypatch.setenv("SURGE_INGEST_DEBUG_WEAK_HASH", "1")
    rng = random.Random(77)
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    seqs, recs, published = {}, [], {}
    for i in range(60000):
        agg = f"agg-{min(i // 12, int(rng.paretovariate(1.1)) % 6000)}"  # new ids keep appearing through the whole topic
        seqs[agg] = seqs.get(agg, 0) + 1
        e = rng.choice([CountIncremented(agg, rng.randrange(1, 9), seqs[agg]), CountDecremented(agg, rng.randrange(1, 9), seqs[agg]), NoOpEvent(agg, seqs[agg])])
        m = fmt.write_event(e)
        recs.append((m.key.encode(), m.value))
        published.setdefault(agg, []).append(e)
    batches = [kw.record_batch(s, recs[s:s + 300], compression="lz4" if (s // 300) % 3 else "none") for s in range(0, len(recs), 300)]
    fetches = [b"".join(batches[i:i + 5]) for i in range(0, len(batches), 5)]
    assert len(fetches) == 40

    def run(pipelined):
        with DeviceDecoder(model.event_json_template()) as d, ReplayEngine(model.event_algebra()) as eng:
            eng.load_csr(np.zeros(1, np.int64), np.zeros(0, dtype=S.EVENT_DTYPE))
            eng.fold()
            if pipelined:
                def push(item):
                    d.push_async(item)
                    return True

                with FramedFetches(fetches, hold=4) as framed, eng.on_own_stream(), PushPipeline(framed, push, 4) as pipe:
                    assert eng.stream_ptr != 0
                    for _ in pipe:
                        d.finish(wait=False)
                        d.fold_into(eng, wait=False)
                        pipe.done()
                    eng.synchronize()
                assert eng.stream_ptr == 0
            else:
                with EventsTopicIngest(frames=True) as g:
                    for f in fetches:
                        g.feed(f)
                        d.push_from(g)
                        d.fold_into(eng)
            n_keys = d.n_keys
            eng.n_agg = n_keys  # (grown inside fold_into)
            return d.keys(), eng.snapshot(), d.stats()

    keys_a, states_a, st_a = run(False)
    keys_b, states_b, st_b = run(True)
    assert keys_a == keys_b and len(keys_a) == len(seqs) and states_a.tobytes() == states_b.tobytes()
    assert st_b["records_delivered"] == 60000 == st_a["records_delivered"] and st_b["pushes"] == 40
    assert (st_b["hash_reseeds"] >= 1) == weak_hash
    by_key = {k: i for i, k in enumerate(keys_b)}
    for agg in list(published)[:: max(1, len(published) // 60)]:  # and against the oracle's fold of the published events, per aggregate
        evs = model.encode_events(published[agg])
        off = np.array([0, evs.shape[0]], np.int64)
        assert states_b[by_key[agg]].tobytes() == oracle.fold_csr(off, evs, None, model.event_algebra())[0].tobytes(), agg


@pytest.mark.gpu
def test_a_batchs_crc32c_is_finished_on_the_device_and_a_damaged_byte_fails_the_push_like_the_host_check():
    """SURGE_INGEST_DEVICE_CRC: the framer checksums the 40 header bytes a batch's CRC-32C covers and
    passes the register on; one wave per section finishes it over the section's bytes on the device (crc_kernel: 16 KiB tiles,
    lane pieces combined by multiplications with x^(8 n) mod P).  Sections of every shape — one record, a few hundred bytes,
    more than one 16 KiB tile, more than two, lz4 frames and plain ones, at whatever alignment the arena gives them: the
    decoded records are the host-checked framer's; one damaged byte — first / middle / last of a section, inside the covered
    header fields, the CRC field itself — fails the PUSH with the host check's status (SURGE_E_CORRUPT), delivers nothing,
    interns nothing, and the decoder takes the undamaged bytes afterwards."""
    from surge_amd.ingest import SECTION_CRC_PENDING

    rng = random.Random(17)
    batches, off = [], 0
    for n, comp in ((1, "none"), (3, "lz4"), (40, "none"), (700, "none"), (2500, "none"), (5000, "lz4"), (17, "none"), (1200, "lz4"), (333, "none"), (1, "lz4")):
        rs = [(f"acct-{rng.randrange(5000):05d}:{off + j}".encode() + b"x" * rng.randrange(0, 3), counter_event(rng.choice([0, 1, 2]), off + j, rng.randrange(-50, 50)), []) for j in range(n)]
        batches.append(kw.record_batch(off, rs, compression=comp))
        off += n
    wire = b"".join(batches)
    starts = np.cumsum([0] + [len(b) for b in batches])

    def decode(data, device_crc):
        with EventsTopicIngest(frames=True, device_lz4=True, device_crc=device_crc) as g, DeviceDecoder(None) as d:
            g.feed(data)
            secs, arena = g.drain_sections()
            assert bool(np.all((secs["codec"] & SECTION_CRC_PENDING) != 0)) == device_crc and secs.shape[0] == len(batches)
            d.push(secs, arena)
            agg, ev, offs, nk = d.result()
            return agg.cpu().numpy(), ev.cpu().numpy(), offs.cpu().numpy(), d.keys()

    host = decode(wire, False)
    dev = decode(wire, True)
    assert dev[3] == host[3] and all(h.tobytes() == g.tobytes() for h, g in zip(host[:3], dev[:3])) and host[0].shape[0] == off
    # damage: (batch, position inside the batch)
    spots = []
    for b in (0, 2, 3, 4, 5, 7, 9):
        L = len(batches[b])
        spots += [(b, 61), (b, L - 1), (b, 61 + (L - 61) // 2)]      # first / last / middle byte of the records section
    spots += [(4, 23), (4, 45), (5, 57), (3, 17), (3, 20)]          # covered header fields (lastOffsetDelta .. recordCount), the CRC field
    for b, at in spots:
        bad = bytearray(wire)
        bad[starts[b] + at] ^= 0x40 if at != 23 else 0x01            # (0x01 of lastOffsetDelta: not the compression bits of the attributes)
        bad = bytes(bad)
        with EventsTopicIngest(frames=True, device_lz4=True) as g:
            with pytest.raises(IngestError) as host_err:
                g.feed(bad)
        assert "CRC-32C mismatch" in str(host_err.value)
        with EventsTopicIngest(frames=True, device_lz4=True, device_crc=True) as g, DeviceDecoder(None) as d:
            g.feed(wire[: starts[1]])                                # a good push first: the table holds a key
            d.push_from(g)
            d.clear()
            keys_before = d.keys()
            with pytest.raises(IngestError) as dev_err:  # (a recordCount no batch of that size can hold is refused by the framer itself)
                g.feed(bad[starts[1]:] if b >= 1 else bad)
                secs, arena = g.drain_sections()
                d.push(secs, arena)
            if "recordCount impossible" in str(dev_err.value):
                assert at == 57 and dev_err.value.status == host_err.value.status
                continue
            # (a damaged LZ4 frame header is already refused by the push's host-side walk of the frame, with the same status)
            assert dev_err.value.status == host_err.value.status, (b, at, str(dev_err.value))
            assert "CRC-32C mismatch (verified on the device)" in str(dev_err.value) or (batches[b][21 + 1] & 7 == 3 and "bad LZ4 frame" in str(dev_err.value)), (b, at, str(dev_err.value))
            assert f"base offset {int(np.cumsum([0] + [x for x, _ in ((1, 0), (3, 0), (40, 0), (700, 0), (2500, 0), (5000, 0), (17, 0), (1200, 0), (333, 0), (1, 0))])[b])}" in str(dev_err.value)
            assert d.keys() == keys_before and d.result()[0].shape[0] == 0
            with EventsTopicIngest(frames=True, device_lz4=True, device_crc=True) as g2:  # ... and the undamaged rest goes through
                g2.feed(wire[starts[1]:])
                d.push_from(g2)
                agg, ev, offs, nk = d.result()
                assert ev.cpu().numpy().tobytes() == host[1][1:].tobytes() and d.keys() == host[3]


@pytest.mark.gpu
def test_parallel_record_chain_is_never_fooled_by_bytes_that_look_like_record_starts():
    """section_kernel finds a batch's records with 64 lanes at once (chain_records_parallel: every lane recognises a record
    start in its own 1/64th of the section, walks from it, and the walks have to link up) and falls back to the one-lane walk
    when they do not.  Topics built to fool the recognition — keys and headers that contain whole fake records (length varint,
    attributes 0, a plausible body) and runs of zero bytes, records of every length from a dozen bytes to a few KiB so that
    starts fall anywhere in a lane's chunk and long records span many chunks, compressed and not — decode exactly like the
    host decoder, record for record; so do thousands of uniform play-json records (the recognition's positive case)."""
    rng = random.Random(23)

    def fake_record(n):
        body = bytes([0, 0, rng.randrange(0, 127) * 2]) + bytes(rng.randrange(256) if rng.random() < 0.7 else 0 for _ in range(n))
        z = len(body) << 1
        return (bytes([z]) if z < 0x80 else bytes([(z & 0x7F) | 0x80, z >> 7])) + body

    def hostile(n):
        kind = rng.randrange(3)
        if kind == 0:
            return b"".join(fake_record(rng.randrange(6, 60)) for _ in range(max(1, n // 40)))
        if kind == 1:
            return bytes(n)
        return bytes(rng.randrange(256) for _ in range(n))

    batches, off = [], 0
    for b in range(200):
        n = rng.randrange(1, 150)
        style = rng.choice(["tiny", "fake-keys", "fake-headers", "long", "mixed"])
        rs = []
        for j in range(n):
            st = style if style != "mixed" else rng.choice(["tiny", "fake-keys", "fake-headers", "long"])
            agg = f"a{rng.randrange(500)}"
            key, hdrs = f"{agg}:{j}".encode(), []
            if st == "fake-keys":
                key = f"{agg}:".encode() + hostile(rng.randrange(8, 80))
            elif st == "fake-headers":
                hdrs = [(b"h", hostile(rng.randrange(8, 120)))]
            elif st == "long":
                hdrs = [(b"blob", hostile(rng.randrange(200, 3000)))]
            rs.append((key, counter_event(rng.choice([0, 1, 2]), off + j, rng.randrange(-5, 5)), hdrs))
        batches.append(kw.record_batch(off, rs, compression=rng.choice(["none", "lz4"])))
        off += n
    wire = b"".join(batches)
    host, host_keys, dev, dev_keys, counters = both_decoders(wire, device_lz4=True)
    assert dev_keys == host_keys and counters["records_delivered"] == host[0].shape[0] == off
    for h, g in zip(host, dev):
        assert h.shape == g.shape and h.tobytes() == g.tobytes()
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    recs = []
    for i in range(30000):
        m = fmt.write_event(CountIncremented(f"agg-{rng.randrange(3000)}", rng.randrange(1000), i + 1))
        recs.append((m.key.encode(), m.value, [(b"trace", hostile(40))] if i % 97 == 0 else []))
    wire = b"".join(kw.record_batch(s, recs[s:s + 140], compression="lz4" if (s // 140) % 3 else "none") for s in range(0, len(recs), 140))
    host, host_keys, dev, dev_keys, _ = both_decoders(wire, model.event_json_template(), device_lz4=True)
    assert dev_keys == host_keys
    for h, g in zip(host, dev):
        assert h.shape == g.shape and h.tobytes() == g.tobytes()
