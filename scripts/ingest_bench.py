#!/usr/bin/env python3
"""N1 throughput on the host: Kafka record batches (v2) -> (aggregate index, 16-byte event, offset) arrays.

One 500-record batch is written once with the test-side wire writer (tests/kafka_wire.py) and replicated with
patched base offsets (the batch CRC does not cover baseOffset), plain and lz4; the C++ decoder
(surge_amd/csrc/ingest.cpp) is timed on feed + drain_fixed16 in fetch-sized chunks, one decoder per partition thread.  No GPU involved.

    python scripts/ingest_bench.py [n_batches=20000] [n_keys=100000] [partition_threads=1]
"""
import ctypes
from concurrent.futures import ThreadPoolExecutor
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import kafka_wire as kw
from surge_amd import _native
from surge_amd.schema import EVENT_DTYPE

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
PER = 500
lib = _native.load()
rng = np.random.default_rng(1)


def run(compression):
    # a handful of distinct batches (distinct key sets) cycled through
    protos = []
    for b in range(16):
        recs = []
        for i in range(PER):
            k = int(rng.integers(0, n_keys))
            ev = np.zeros(1, dtype=EVENT_DTYPE)
            ev["type"], ev["seq"], ev["raw"] = 1 + (i & 1), i + 1, np.uint64(int(rng.integers(1, 10)))
            recs.append((f"acct-{k:08d}:{i}".encode(), ev.tobytes()))
        protos.append(bytearray(kw.record_batch(0, recs, compression=compression)))
    parts = []
    for b in range(n_batches):
        p = bytearray(protos[b % len(protos)])
        struct.pack_into(">q", p, 0, b * PER)
        parts.append(bytes(p))
    # fetch-sized chunks of whole batches (~1 MiB, like max.partition.fetch.bytes), drained after every feed
    chunks, cur, cur_len = [], [], 0
    for part in parts:
        cur.append(part)
        cur_len += len(part)
        if cur_len >= (1 << 20):
            chunks.append(b"".join(cur))
            cur, cur_len = [], 0
    if cur:
        chunks.append(b"".join(cur))
    arrs = [(ctypes.c_uint8 * len(c)).from_buffer_copy(c) for c in chunks]
    wire = sum(len(c) for c in chunks)
    n = n_batches * PER
    agg = np.zeros(n, dtype=np.int64); ev = np.zeros(n, dtype=EVENT_DTYPE); off = np.zeros(n, dtype=np.int64)
    vp = ctypes.c_void_p
    def one_partition(out):
        agg, ev, off = out
        h = ctypes.c_void_p()
        assert lib.surge_ingest_create(1, ctypes.byref(h)) == 0
        consumed, got = ctypes.c_int64(0), ctypes.c_int64(0)
        done = 0
        for a in arrs:
            rc = lib.surge_ingest_feed(h, a, len(a), ctypes.byref(consumed))
            rc2 = lib.surge_ingest_drain_fixed16(h, n - done, vp(agg.ctypes.data + 8 * done), vp(ev.ctypes.data + 16 * done),
                                                 vp(off.ctypes.data + 8 * done), ctypes.byref(got))
            assert rc == 0 and rc2 == 0 and consumed.value == len(a), (rc, rc2)
            done += got.value
        lib.surge_ingest_destroy(h)
        assert done == n and off[-1] == n - 1 and int(agg.max()) < n_keys

    # one decoder per partition thread (decoders share nothing); ctypes releases the GIL during the calls
    outs = [(agg, ev, off)] + [(np.zeros_like(agg), np.zeros_like(ev), np.zeros_like(off)) for _ in range(n_threads - 1)]
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        if n_threads == 1:
            one_partition(outs[0])
        else:
            with ThreadPoolExecutor(max_workers=n_threads) as pool:
                list(pool.map(one_partition, outs))
        best = min(best, time.perf_counter() - t0)
    n_total = n * n_threads
    return {"compression": compression, "partitions": n_threads, "records": n_total, "wire_MB": wire * n_threads / 1e6, "seconds": best,
            "records_per_sec": n_total / best, "wire_MBps": wire * n_threads / 1e6 / best}


print(json.dumps({"threads": n_threads, "records_per_batch": PER, "distinct_keys": n_keys, "runs": [run("none"), run("lz4")]}))
