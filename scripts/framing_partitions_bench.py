#!/usr/bin/env python3
"""How fast do the host threads frame a fetch response of 64 partitions (surge_ingest_feed_drain_many)?  No GPU work:
the consumer drops the parts.  python scripts/framing_partitions_bench.py [records_per_fetch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import topic_gen  # noqa: E402
from surge_amd.ingest import PartitionedFramedFetches  # noqa: E402
from surge_amd.snapshot import RecordBatchWriter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P, F = 64, 12
rng = np.random.default_rng(0)
fetches = []
with RecordBatchWriter(P, 0, 16384, "lz4") as w:
    for f in range(F):
        agg = rng.integers(0, 10**7, n)
        k, ko, v, vo = topic_gen.counter_records(agg, rng.integers(0, 3, n), rng.integers(0, 1000, n), np.full(n, f + 1))
        fetches.append(topic_gen.frame_partitions(w, (agg % P).astype(np.int32), k, ko, v, vo))
wire = sum(len(x) for x in fetches[0] if x)
print(f"{n} records per fetch, {wire / n:.1f} wire bytes per record, {F} fetches")
for pageable in (0, 1):
    os.environ["SURGE_INGEST_PAGEABLE_ARENA"] = str(pageable)
    for threads in (1, 2, 4, 8, 16):
        with PartitionedFramedFetches(iter(fetches), P, threads=threads, hold=3, overlap=False) as framed:
            t0 = time.perf_counter()
            for parts in framed:
                pass
            dt = time.perf_counter() - t0
            ms = [x * 1e3 for x in framed.framing_seconds]
        print(f"arena {'pageable' if pageable else 'page-locked'}  threads {threads:2d}: {np.median(ms[4:]):7.3f} ms per fetch (median of the last {F - 4}), first four {[round(x, 1) for x in ms[:4]]}, "
              f"{n * F / dt / 1e6:.0f} M records/s")
