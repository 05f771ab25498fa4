#!/usr/bin/env python3
"""Events-topic bytes -> (aggregate index, 16-byte event) arrays: the host decoder (one partition thread) against host framing
+ the device decoder, on the same wire bytes.  Counter fixture events as play-json text, and the same records with 16-byte
values; uncompressed and LZ4 batches of 16 KiB (the reference producer's batch size).   python scripts/ingest_gpu_bench.py [records]   (needs a GPU)"""
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
import torch

import kafka_wire as kw
from fixture_models import CounterBusinessLogic, CountDecremented, CountIncremented, NoOpEvent
from surge_amd.ingest import DeviceDecoder, EventsTopicIngest

n_records = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
# records per batch: what the reference's producer packs into one batch — it closes a batch at 16 KiB of records
# (kafka.publisher.batch-size = 16384, reference.conf:115): ~140 play-json Counter events, ~400 16-byte ones
PER_BY_MODE = {"json": 140, "fixed16": 400}
rng = np.random.default_rng(1)
bl = CounterBusinessLogic()
model, fmt = bl.command_model(), bl.event_write_formatting()
tmpl = model.event_json_template()
res = {"records_per_case": n_records, "batch_records": PER_BY_MODE}
for mode in ("json", "fixed16"):
    PER = PER_BY_MODE[mode]
    n_batches = n_records // PER
    for codec in ("none", "lz4"):
        protos = []
        for b in range(16):
            recs = []
            for i in range(PER):
                agg = f"acct-{int(rng.integers(0, 100000)):08d}"
                e = [CountIncremented(agg, int(rng.integers(0, 1000)), i + 1), CountDecremented(agg, int(rng.integers(0, 1000)), i + 1), NoOpEvent(agg, i + 1)][i % 3]
                m = fmt.write_event(e)
                recs.append((m.key.encode(), m.value if mode == "json" else model.encode_events([e]).tobytes()))
            protos.append(bytearray(kw.record_batch(0, recs, compression=codec)))
        parts = []
        for b in range(n_batches):
            p = bytearray(protos[b % len(protos)])
            struct.pack_into(">q", p, 0, b * PER)  # baseOffset is outside the CRC
            parts.append(bytes(p))
        wire = b"".join(parts)
        n = n_batches * PER
        with EventsTopicIngest() as g:
            t0 = time.perf_counter()
            g.feed(wire)
            t1 = time.perf_counter()
            h_agg, h_ev, h_off = g.drain_json(tmpl) if mode == "json" else g.drain_fixed16()
            t2 = time.perf_counter()
        r = {"wire_bytes": len(wire), "host_decoder": {"feed_s": t1 - t0, "drain_s": t2 - t1, "records_per_sec": n / (t2 - t0)}}
        variants = [("framing_plus_device_decoder", False)] + ([("framing_plus_device_decoder_lz4_on_device", True)] if codec == "lz4" else [])
        for label, device_lz4 in variants:
          with EventsTopicIngest(frames=True, device_lz4=device_lz4) as g, DeviceDecoder(tmpl if mode == "json" else None) as d:
            for rep in range(3):  # the last pass runs with warm buffers (both of the framer's arenas) and a populated key table
                d.clear()
                t0 = time.perf_counter()
                g.feed(wire)
                t1 = time.perf_counter()
                sections, arena = g.drain_sections()
                t2 = time.perf_counter()
                d.push(sections, arena)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
            agg, ev, off, n_keys = d.result()
            same = bool((agg.cpu().numpy() == h_agg).all() and (ev.cpu().numpy().view(h_ev.dtype).reshape(-1) == h_ev).all() and (off.cpu().numpy() == h_off).all())
          r[label] = {"host_framing_s": t1 - t0, "drain_sections_s": t2 - t1, "device_push_s": t3 - t2,
                      "records_per_sec": n / (t3 - t0), "device_push_records_per_sec": n / (t3 - t2), "equal_to_host_decoder": same, "keys": n_keys}
        r["speedup_one_host_thread"] = max(r[l]["records_per_sec"] for l, _ in variants) / r["host_decoder"]["records_per_sec"]
        res[f"{mode}/{codec}"] = r
print(json.dumps(res))
