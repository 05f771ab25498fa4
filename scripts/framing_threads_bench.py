#!/usr/bin/env python3
"""Host framing of one 1 M-record fetch (16 KiB batches of play-json Counter events, as bench.py --workload e2e feeds them)
with 1 .. 8 CRC threads (surge_ingest_set_threads): milliseconds per fetch and wire GB/s.  CPU only."""
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

PER, BATCHES = 140, 7142
rng = np.random.default_rng(1)
fmt = CounterBusinessLogic().event_write_formatting()
res = {"host_cpus": os.cpu_count()}
for codec in ("lz4", "none"):
    protos = []
    for b in range(64):
        recs = []
        for i in range(PER):
            agg = f"acct-{int(rng.integers(0, 1_000_000)):08d}"
            e = [CountIncremented(agg, int(rng.integers(0, 1000)), i + 1), CountDecremented(agg, int(rng.integers(0, 1000)), i + 1), NoOpEvent(agg, i + 1)][i % 3]
            m = fmt.write_event(e)
            recs.append((m.key.encode(), m.value))
        protos.append(bytearray(kw.record_batch(0, recs, compression=codec)))
    parts = []
    for b in range(BATCHES):
        p = bytearray(protos[b % len(protos)])
        struct.pack_into(">q", p, 0, b * PER)
        parts.append(bytes(p))
    wire = b"".join(parts)
    row = {"wire_bytes": len(wire)}
    for threads in (1, 2, 4, 8):
        with EventsTopicIngest(frames=True, device_lz4=True, threads=threads) as g:
            best = 1e9
            for rep in range(6):
                t0 = time.perf_counter()
                g.feed(wire)
                sections, _ = g.drain_sections()
                best = min(best, time.perf_counter() - t0)
            assert sections.shape[0] == BATCHES
        row[f"threads_{threads}"] = {"ms_per_fetch": round(best * 1e3, 3), "wire_GBps": round(len(wire) / best / 1e9, 2)}
    res[codec] = row
print(json.dumps(res))
