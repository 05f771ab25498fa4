"""Events-topic record values as the reference's plugins write them (JSON text) -> 16-byte fixed events, decoded in the
library (surge_event_json_decode / surge_ingest_drain_json) — the mirror image of the GPU state encoders.

Pinned on: Python's json module (a third party to the decoder) for the text, ``float(repr)`` / IEEE for the doubles, the
model's own ``encode_event`` for the fixed event.  play-json's exact text is parity-unpinned: the tests
therefore feed every variation its writer could produce — any field order, discriminator first or last, extra fields,
whitespace, exponents — and require the same event."""
import json
import random
import uuid

import numpy as np
import pytest

from fixture_models import (
    BA_CREATED,
    BA_UPDATED,
    BankAccountCommandModel,
    BankAccountCreated,
    BankAccountUpdated,
    CounterBusinessLogic,
    CountDecremented,
    CountIncremented,
    NoOpEvent,
)
from surge_amd import schema as S
from surge_amd.ingest import ARG_F64, ARG_I32, ARG_NONE, EventJsonTemplate, EventsTopicIngest, IngestError


def bank_event_json(evt, rng, fq=True):
    """What ``Json.toJson(evt)(Json.format[BankAccountEvent])`` writes, up to what play-json leaves open."""
    if isinstance(evt, BankAccountCreated):
        body = {"accountNumber": str(evt.accountNumber), "accountOwner": evt.accountOwner, "securityCode": evt.securityCode, "balance": evt.balance}
        name = "docs.command.BankAccountCreated"
    else:
        body = {"accountNumber": str(evt.accountNumber), "newBalance": evt.newBalance}
        name = "docs.command.BankAccountUpdated"
    items = list(body.items())
    items.insert(rng.choice([0, len(items)]), ("_type", name))  # discriminator first or last
    return json.dumps(dict(items), separators=(",", ":")).encode()


def test_counter_events_decode_to_what_encode_event_builds():
    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    tmpl = model.event_json_template()
    rng = random.Random(1)
    for _ in range(500):
        seq = rng.randrange(0, 2**31)
        k = rng.choice([0, 1, -1, 7, 2**31 - 1, -(2**31), rng.randrange(-(2**31), 2**31)])
        evt = rng.choice([CountIncremented("agg-1", k, seq), CountDecremented("x:y", k, seq), NoOpEvent("n", seq)])
        value = fmt.write_event(evt).value
        assert tmpl.decode(value).tobytes() == model.encode_events([evt])[0].tobytes()
        # the same object with the fields shuffled, extra fields, nested values and whitespace
        o = json.loads(value)
        o["extra"] = {"nested": [1, 2, {"_type": "countDecremented"}], "s": 'a"b}'}
        o["flag"], o["nothing"] = True, None
        items = list(o.items())
        rng.shuffle(items)
        noisy = json.dumps(dict(items), indent=rng.choice([None, 1])).encode()
        assert tmpl.decode(noisy).tobytes() == model.encode_events([evt])[0].tobytes()


def test_bank_account_doubles_are_correctly_rounded():
    model = BankAccountCommandModel()
    tmpl = model.event_json_template()
    rng = random.Random(2)
    acct = uuid.UUID(int=7)
    values = [0.0, -0.0, 1000.0, 1100.0, 0.1, 1e-320, 5e-324, 1.7976931348623157e308, 123456789.125, 1 / 3]
    values += [rng.uniform(-1e9, 1e9) for _ in range(300)] + [rng.random() * 10.0 ** rng.randrange(-300, 300) for _ in range(200)]
    for v in values:
        for evt in (BankAccountCreated(acct, "Jane", "1234", v), BankAccountUpdated(acct, v)):
            got = tmpl.decode(bank_event_json(evt, rng))
            assert int(got["type"]) == (BA_CREATED if isinstance(evt, BankAccountCreated) else BA_UPDATED)
            assert got["raw"] == np.float64(v).view(np.uint64), v  # bit-exact, also for -0.0 and subnormals
    # other spellings of the same number (play-json / BigDecimal may write 1.1E+3 or 1100): same bits
    for text, v in (("1.1E+3", 1100.0), ("1100", 1100.0), ("1100.000", 1100.0), ("-2.5e-3", -0.0025), ("1E400", float("inf"))):
        raw = b'{"_type":"docs.command.BankAccountUpdated","accountNumber":"x","newBalance":' + text.encode() + b"}"
        assert tmpl.decode(raw)["raw"] == np.float64(v).view(np.uint64)


def test_what_the_template_does_not_describe_is_refused_with_a_reason():
    tmpl = CounterBusinessLogic().command_model().event_json_template()
    bad = {
        b"null": "not a JSON object",                                     # ExceptionThrowingEvent's writes = JsNull (:45)
        b"": "not a JSON object",
        b'{"_type":"mystery","sequenceNumber":1}': "unknown event type",
        b'{"sequenceNumber":1,"incrementBy":1}': 'no string field "_type"',
        b'{"_type":"countIncremented","sequenceNumber":1}': 'no numeric field "incrementBy"',
        b'{"_type":"countIncremented","sequenceNumber":1,"incrementBy":1.5}': "is not an Int",
        b'{"_type":"countIncremented","sequenceNumber":1,"incrementBy":2147483648}': "is not an Int",
        b'{"_type":"countIncremented","sequenceNumber":"1","incrementBy":1}': 'no numeric field "sequenceNumber"',
        b'{"_type":"no-op","sequenceNumber":1} trailing': "trailing bytes",
        b'{"_type":"no-op","sequenceNumber":1': "expected ',' or '}'",
        b'{"_type":"no-op" "sequenceNumber":1}': "expected ',' or '}'",
        b'{"_type":"no-op,"sequenceNumber":1}': "",
    }
    for value, why in bad.items():
        with pytest.raises(IngestError) as ei:
            tmpl.decode(value)
        assert ei.value.status == -7 and why in str(ei.value), (value, str(ei.value))
    # a template is validated before use
    with pytest.raises(IngestError):
        EventJsonTemplate("", [("a", 0, "", "", ARG_NONE), ("b", 1, "", "", ARG_NONE)]).decode(b"{}")  # two types, no discriminator
    assert int(EventJsonTemplate("", [("", 3, "n", "v", ARG_I32)]).decode(b'{"v":-5,"n":9}')["type"]) == 3  # single-class topic


def test_drain_json_decodes_a_whole_topic_without_per_record_python():
    """Record batches (lz4, transactions, an aborted flush, the producer's flush record) whose values are the Counter
    fixture's JSON events -> (agg_idx, events) arrays in one library call."""
    import kafka_wire as kw

    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    rng = random.Random(3)
    wire, off, expect = [kw.record_batch(0, [(b"", b"")])], 1, []  # the flush record first (KafkaProducerActorImpl.scala:322)
    for flush in range(200):
        events = []
        for _ in range(rng.randrange(1, 12)):
            agg = f"agg-{rng.randrange(40)}"
            events.append(rng.choice([CountIncremented(agg, rng.randrange(100), flush), CountDecremented(agg, rng.randrange(100), flush), NoOpEvent(agg, flush)]))
        outcome = kw.COMMIT if rng.random() < 0.9 else kw.ABORT
        msgs = [fmt.write_event(e) for e in events]
        wire.append(kw.record_batch(off, [(m.key.encode(), m.value) for m in msgs], compression=rng.choice(["lz4", "none"]), transactional=True, producer_id=9))
        off += len(msgs)
        wire.append(kw.control_batch(off, 9, outcome))
        off += 1
        if outcome == kw.COMMIT:
            expect += events
    with EventsTopicIngest() as g:
        g.feed(b"".join(wire))
        agg_idx, events, offsets = g.drain_json(model.event_json_template())
        keys = g.key_table()
    assert events.tobytes() == model.encode_events(expect).tobytes()
    assert [keys.keys[i] for i in agg_idx] == [e.aggregateId for e in expect]
    assert list(offsets) == sorted(offsets)
    # a value that does not decode: nothing is popped, the error names the record
    with EventsTopicIngest() as g:
        g.feed(kw.record_batch(0, [(b"a:1", fmt.write_event(CountIncremented("a", 1, 1)).value), (b"a:2", b"null")]))
        with pytest.raises(IngestError) as ei:
            g.drain_json(model.event_json_template())
        assert "offset 1" in str(ei.value) and "not a JSON object" in str(ei.value) and g.ready == 2


@pytest.mark.gpu
def test_recovery_from_a_play_json_events_topic_runs_without_per_record_python(monkeypatch):
    """R12 end to end on a real Surge events topic: JSON event values -> library decoder -> device group-by (the K3 radix
    path) -> fold -> getAggregateBytes, checked against the literal handle_event fold.  The plugin's per-record reader and
    the host-side group-by must not run at all."""
    import time

    import kafka_wire as kw
    import surge_amd.log as log_mod
    from surge_amd.store import GpuReplayStateStore

    bl = CounterBusinessLogic()
    model, fmt = bl.command_model(), bl.event_write_formatting()
    rng = random.Random(4)
    n_agg, expect, wire, off = 3000, {}, [], 0
    for flush in range(400):
        events = []
        for _ in range(50):
            agg = f"acct-{rng.randrange(n_agg):08d}"
            seq = (expect[agg].version if agg in expect and expect[agg] is not None else 0) + 1
            e = rng.choice([CountIncremented(agg, rng.randrange(1000), seq), CountDecremented(agg, rng.randrange(1000), seq), NoOpEvent(agg, seq)])
            events.append(e)
            expect[agg] = model.handle_event(expect.get(agg), e)
        msgs = [fmt.write_event(e) for e in events]
        wire.append(kw.record_batch(off, [(m.key.encode(), m.value) for m in msgs], compression="lz4", transactional=True, producer_id=3))
        off += len(msgs)
        wire.append(kw.control_batch(off, 3, kw.COMMIT))
        off += 1
    topic = b"".join(wire)

    def forbidden(*a, **k):
        raise AssertionError("per-record Python on the recovery path")

    monkeypatch.setattr(type(fmt), "read_event", forbidden)
    monkeypatch.setattr(type(model), "encode_event", forbidden)
    monkeypatch.setattr(log_mod, "group_by_aggregate", forbidden)
    store = GpuReplayStateStore(bl)
    try:
        t0 = time.perf_counter()
        counters = store.restore_from_topic(topic)
        dt = time.perf_counter() - t0
        assert counters["records_delivered"] == 400 * 50
        print(f"recovered {counters['records_delivered']} JSON events of {len(expect)} aggregates in {dt * 1e3:.1f} ms ({counters['records_delivered'] / dt:.3g} records/s)")
        wf = bl.aggregate_write_formatting()
        for k, st in expect.items():
            assert store.get_aggregate_bytes(k) == wf.write_state(st).value, k
    finally:
        store.close()
