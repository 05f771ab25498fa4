#!/usr/bin/env python3
"""Host throughput of the JSON event-value decoder (surge_ingest_drain_json): record batches whose values are the Counter
fixture's play-json event text -> (aggregate index, 16-byte event, offset) arrays, next to the same records with fixed-16
values through surge_ingest_drain_fixed16.  One partition thread, no GPU.   python scripts/event_decode_bench.py [n_batches]"""
import ctypes
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np

import kafka_wire as kw
from fixture_models import CounterBusinessLogic, CountDecremented, CountIncremented, NoOpEvent
from surge_amd.ingest import EventsTopicIngest

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
PER = 500
rng = np.random.default_rng(1)
bl = CounterBusinessLogic()
model, fmt = bl.command_model(), bl.event_write_formatting()
tmpl = model.event_json_template()
res = {}
for mode in ("json", "fixed16"):
    protos = []
    for b in range(8):
        recs = []
        for i in range(PER):
            agg = f"acct-{int(rng.integers(0, 100000)):08d}"
            e = [CountIncremented(agg, int(rng.integers(0, 1000)), i + 1), CountDecremented(agg, int(rng.integers(0, 1000)), i + 1), NoOpEvent(agg, i + 1)][i % 3]
            m = fmt.write_event(e)
            recs.append((m.key.encode(), m.value if mode == "json" else model.encode_events([e]).tobytes()))
        protos.append(bytearray(kw.record_batch(0, recs)))
    parts = []
    for b in range(n_batches):
        p = bytearray(protos[b % len(protos)])
        struct.pack_into(">q", p, 0, b * PER)
        parts.append(bytes(p))
    wire = b"".join(parts)
    with EventsTopicIngest() as g:
        t0 = time.perf_counter()
        g.feed(wire)
        t1 = time.perf_counter()
        agg_idx, events, offsets = g.drain_json(tmpl) if mode == "json" else g.drain_fixed16()
        t2 = time.perf_counter()
    n = n_batches * PER
    assert events.shape[0] == n
    res[mode] = {"records": n, "wire_bytes": len(wire), "feed_s": t1 - t0, "drain_s": t2 - t1, "records_per_sec": n / (t2 - t0),
                 "drain_ns_per_record": (t2 - t1) / n * 1e9}
print(json.dumps(res))
