#!/usr/bin/env python3
"""The in-process exchange (surge_replay_allgather) timed on whatever GPUs one process sees: every handle folds one
shard of config C4 (the 10 M-aggregate Zipf log sharded by Kafka partition over --shards ranks; --handles of them are
materialised), then all handles exchange their final snapshots with peer copies.

On a one-GPU box the handles share the device, so folds serialise and the "peer" copies are local HBM copies: what the
numbers show there is the asynchrony (does a step that exchanges cost more than a step that only folds?), not xGMI.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from surge_amd import synth
from surge_amd.dist import local_aggregate_ids
from surge_amd.replay import ReplayEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aggregates", type=int, default=10_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--handles", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    ndev = torch.cuda.device_count()
    engines, n_events = [], 0
    for r in range(args.handles):
        dev = torch.device("cuda", r % ndev)
        eng = ReplayEngine(device=r % ndev)
        ids = local_aggregate_ids(args.aggregates, 64, r, args.shards, dev, eng)
        lens = synth.zipf_lengths(ids, 3)
        so, ev = synth.csr_log_device(lens, 3 + r)  # the shard's own events (contents do not matter for timing)
        n_events += int(so[-1].item())
        eng.load_csr(so, ev)
        eng.fold()
        engines.append(eng)

    def sync():
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize()

    def loop(exchange):
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            for e in engines:
                e.fold()
            if exchange:
                ReplayEngine.allgather_group(engines, slot=i & 1)
        sync()
        return (time.perf_counter() - t0) / args.steps * 1e3

    loop(True)  # warm-up: allocations, communicator objects
    fold_only = min(loop(False) for _ in range(3))
    with_exchange = min(loop(True) for _ in range(3))
    alone = []
    for i in range(10):
        sync()
        t0 = time.perf_counter()
        ReplayEngine.allgather_group(engines, slot=i & 1)
        sync()
        alone.append((time.perf_counter() - t0) * 1e3)
    counts, mx = engines[0].comm_counts(0)
    print(json.dumps({
        "what": f"{args.handles} handles in one process on {min(ndev, args.handles)} GPU(s), each holding one of {args.shards} shards of the "
                f"{args.aggregates}-aggregate Zipf log; surge_replay_allgather after every fold",
        "states_per_handle": [int(c) for c in counts], "events_folded_per_step": n_events,
        "ms_per_step_fold_only": fold_only, "ms_per_step_fold_and_exchange": with_exchange,
        "exchange_alone_ms_median": float(np.median(alone)),
        "exchange_hidden_fraction": max(0.0, min(1.0, 1.0 - (with_exchange - fold_only) / float(np.median(alone)))),
        "wire_bytes_per_exchange": int(sum(counts)) * 40 * args.handles,
        "note": "handles share one device here: copies are local, folds serialise" if ndev < args.handles else "one handle per device: copies cross xGMI",
    }))
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
