#!/usr/bin/env python3
"""Config C5: streaming micro-batches onto a GPU-resident state store.

  * population: A aggregates resident in HBM (default 10 M x 64 B = 640 MB), first recovered by a full fold
  * ingest: 600 micro-batches of B events (default 100 000 = 60 s of a 1 M events/s stream in 100 ms batches) whose
    aggregate ids are Zipf-popular, in topic order; each batch goes through surge_replay_append_events: pinned staging
    + H2D, DEVICE group-by (stable radix sort + head scan, stream_kernels.hip), fold onto the resident state (K3)
  * every S batches (default 30 = 3 s, mirrors kafka.streams.commit-interval-ms=3000,
    modules/common/src/main/resources/reference.conf:19) the state-topic delta is published: delta kernel -> filtered GPU
    JSON encoder -> D2H -> Kafka record batches (BulkSnapshotPublisher) — the incremental KTable snapshot.

Prints one JSON line: sustained ingest capacity (events/s), batch latency p50/p99/max (host wall clock: staging + H2D +
group-by + kernel + sync), kernel-only time, snapshot time.  Latency-bound, not bandwidth-bound.  (Parity of this exact
path — successive append_events batches and delta publishes against the CPU oracle — is tests/test_store.py and
tests/test_gpu_parity.py; a benchmark script does not touch oracle/.)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aggregates", type=int, default=10_000_000)
    ap.add_argument("--batch-events", type=int, default=100_000)
    ap.add_argument("--batches", type=int, default=600)
    ap.add_argument("--snapshot-every", type=int, default=30, help="publish a state-topic delta every N batches (0 = never)")
    ap.add_argument("--device-batches", action="store_true", help="batches already in HBM (no staging / H2D)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from surge_amd import synth
    from surge_amd.dist import ID_DIGITS, ID_PREFIX, id_table_utf16
    from surge_amd.replay import ReplayEngine
    from surge_amd.snapshot import BulkSnapshotPublisher

    dev = torch.device("cuda:0")
    A, B = args.aggregates, args.batch_events
    # initial recovery: a short uniform log (16 events per aggregate) folded by the rows kernel
    so, ev = synth.fixed_log_device(A, 16, 5, dev, mix=synth.C1_MIX)
    eng = ReplayEngine()
    eng.load_csr(so, ev)
    eng.fold()
    eng.synchronize()
    # key table acct-%08d without Python strings: UTF-16 for the partitioner, the same code units as bytes for the encoder
    ids = torch.arange(A, dtype=torch.int64, device=dev)
    u16, o16 = id_table_utf16(ids)
    pub = BulkSnapshotPublisher(eng, None, 64, tables=(u16.to(torch.uint8), o16.clone(), u16, o16))
    t0 = time.perf_counter()
    pub.publish()  # the full snapshot after recovery (baseline for the deltas)
    full_snapshot_s = time.perf_counter() - t0
    full_t = dict(pub.timings)

    rng = np.random.default_rng(7)
    cdf = synth.zipf_cdf(4096)
    lat, kern, snap_ms, touched, snap_bytes = [], [], [], [], []
    t_all0 = time.perf_counter()
    for b in range(args.batches):
        # Zipf-popular aggregate ids (rank -> id through a fixed permutation-free mapping: id = rank * 2654435761 mod A)
        ranks = np.searchsorted(cdf, rng.random(B)).astype(np.int64) * (A // 4096) + rng.integers(0, max(A // 4096, 1), B)
        agg_idx = (ranks * 2654435761) % A
        words = synth.event_words(np.arange(B, dtype=np.int64) + b * B, agg_idx, np.arange(B, dtype=np.int64), 11, synth.C1_MIX)
        events = synth.to_event_records(words)
        if args.device_batches:
            d_idx, d_ev = torch.from_numpy(agg_idx).to(dev), torch.from_numpy(words).to(dev)
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if args.device_batches:
            eng.append_events(d_idx, d_ev)
        else:
            eng.append_events(agg_idx, events)   # pinned staging + H2D + device group-by + K3
        eng.synchronize()
        t1 = time.perf_counter()
        lat.append((t1 - t0) * 1e3)
        kern.append(eng.stats().last_fold_kernel_ms)
        if args.snapshot_every > 0 and (b + 1) % args.snapshot_every == 0:
            t0 = time.perf_counter()
            batches = pub.publish()
            snap_ms.append((time.perf_counter() - t0) * 1e3)
            touched.append(int(pub.timings["values"] + pub.timings["tombstones"]))
            snap_bytes.append(sum(len(x) for x in batches.values()))
    total_s = time.perf_counter() - t_all0
    lat = np.array(lat)
    print(json.dumps({
        "workload": f"C5: {A} resident aggregates, {args.batches} micro-batches x {B} events "
                    f"({'device-resident' if args.device_batches else 'host'} batches), state-topic delta every {args.snapshot_every}",
        "sustained_events_per_sec": B * args.batches / float(lat.sum() / 1e3 + sum(snap_ms) / 1e3),
        "ingest_only_events_per_sec": B * args.batches / float(lat.sum() / 1e3),
        "target_ingest_events_per_sec": 1_000_000,
        "batch_latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
        "kernel_ms_per_batch": float(np.mean(kern)),
        "snapshot_ms": {"mean": float(np.mean(snap_ms)) if snap_ms else None, "max": float(np.max(snap_ms)) if snap_ms else None, "n": len(snap_ms)},
        "snapshot_published_aggregates_mean": float(np.mean(touched)) if touched else None,
        "snapshot_record_batch_bytes_mean": float(np.mean(snap_bytes)) if snap_bytes else None,
        "full_snapshot_after_recovery": {"seconds": full_snapshot_s, **full_t},
        "wall_s_including_event_generation": total_s,
    }))
    pub.close()
    eng.close()


if __name__ == "__main__":
    main()
