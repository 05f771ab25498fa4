#!/usr/bin/env python3
"""N2 x N3 timing: a full and an incremental state-topic snapshot of a large resident state, GPU delta + encode, D2H,
RecordBatch v2 encoding (scripts/snapshot_bench.py [aggregates]); prints one JSON line (profiles/r02_snapshot_n2.json)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from surge_amd import synth
from surge_amd.replay import ReplayEngine
from surge_amd.snapshot import BulkSnapshotPublisher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
compression = sys.argv[2] if len(sys.argv) > 2 else "none"  # "lz4": compress the record batches like the reference producer
dev = torch.device("cuda:0")
lens = synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3, max_len=64)
so, ev = synth.csr_log_device(lens, 3, mix=synth.C1_MIX)
keys = [f"acct-{i:08d}" for i in range(n)]
with ReplayEngine() as eng:
    eng.load_csr(so, ev)
    eng.fold()
    t0 = time.perf_counter()
    pub = BulkSnapshotPublisher(eng, keys, 64, compression=compression)
    setup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    full = pub.publish()
    full_s = time.perf_counter() - t0
    full_t = dict(pub.timings)
    # touch 1 % of the aggregates, publish the delta
    m = n // 100
    idx = torch.randperm(n, device=dev)[:m].to(torch.int64)
    be = synth.to_event_records(synth.event_words(torch.arange(m, device=dev), idx, torch.arange(m, device=dev), 9, synth.C1_MIX))
    eng.append_events(idx.cpu().numpy(), be)
    t0 = time.perf_counter()
    delta = pub.publish()
    delta_s = time.perf_counter() - t0
    print(json.dumps({
        "aggregates": n, "partitions": 64, "key_table_setup_s": setup_s,
        "full_snapshot": {"seconds": full_s, "record_batch_bytes": sum(len(b) for b in full.values()), **full_t,
                          "aggregates_per_sec": n / full_s},
        "incremental_1pct": {"seconds": delta_s, "record_batch_bytes": sum(len(b) for b in delta.values()), **pub.timings},
    }))
    pub.close()
