#!/usr/bin/env python3
"""Does host framing + device decode scale over partition threads?  T threads, each with its own partition: its own
EventsTopicIngest (frames, LZ4 left to the device), its own surge_device_decoder on its own HIP stream — what a node that
owns T partitions of the events topic runs (partitions are independent state stores in Surge: KafkaPartitioner.scala:8).
Every thread decodes the same number of fetches of play-json Counter events; records/s over all threads.
python scripts/ingest_gpu_mt_bench.py [fetches_per_thread]   (needs a GPU)"""
import json
import os
import struct
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np
import pyarrow as pa
import torch

import kafka_wire as kw
from fixture_models import CounterBusinessLogic, CountDecremented, CountIncremented, NoOpEvent
from surge_amd.ingest import DeviceDecoder, EventsTopicIngest

n_fetch_per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 6
PER, BATCHES = 140, 7142  # a fetch = 1 M records in 16 KiB batches (kafka.publisher.batch-size = 16384, reference.conf:115)
bl = CounterBusinessLogic()
model, fmt = bl.command_model(), bl.event_write_formatting()
tmpl = model.event_json_template()
lz4 = lambda raw: pa.Codec("lz4").compress(raw, asbytes=True)  # noqa: E731  (liblz4 frames, as kafka-clients' look)


def partition_wire(part, rng):
    protos = []
    for b in range(16):
        recs = []
        for i in range(PER):
            agg = f"p{part}-acct-{int(rng.integers(0, 200000)):08d}"
            e = [CountIncremented(agg, int(rng.integers(0, 1000)), i + 1), CountDecremented(agg, int(rng.integers(0, 1000)), i + 1), NoOpEvent(agg, i + 1)][i % 3]
            m = fmt.write_event(e)
            recs.append((m.key.encode(), m.value))
        protos.append(bytearray(kw.record_batch(0, recs, compression="lz4", compressor=lz4)))
    parts = []
    for b in range(BATCHES):
        p = bytearray(protos[b % len(protos)])
        struct.pack_into(">q", p, 0, b * PER)
        parts.append(bytes(p))
    return b"".join(parts)


res = {"fetch_records": PER * BATCHES, "fetches_per_thread": n_fetch_per_thread, "threads": {}}
for T in (1, 2, 4, 8, 16):
    rng = np.random.default_rng(T)
    wires = [partition_wire(p, rng) for p in range(T)]
    streams = [torch.cuda.Stream() for _ in range(T)]
    ingests = [EventsTopicIngest(frames=True, device_lz4=True) for _ in range(T)]
    decoders = [DeviceDecoder(tmpl, stream=ctypes_stream) for ctypes_stream in [__import__("ctypes").c_void_p(s.cuda_stream) for s in streams]]
    delivered = [0] * T

    def work(t):
        g, d = ingests[t], decoders[t]
        for _ in range(n_fetch_per_thread):
            g.feed(wires[t])
            d.push_from(g)
            delivered[t] += d.result()[0].shape[0]
            d.clear()

    for t in range(T):  # warm-up: buffers, pinned arenas, key tables
        work(t)
    delivered = [0] * T
    torch.cuda.synchronize()
    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(delivered)
    res["threads"][T] = {"records": n, "seconds": dt, "records_per_sec": n / dt, "wire_GBps": sum(len(w) for w in wires) * n_fetch_per_thread / dt / 1e9}
    print(f"T={T}: {n / dt / 1e6:.1f} M records/s", file=sys.stderr, flush=True)
    for d in decoders:
        d.close()
    for g in ingests:
        g.close()
print(json.dumps(res))
