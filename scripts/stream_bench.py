#!/usr/bin/env python3
"""Config C5 from the command line: `python bench.py --workload c5` with this script's historical flag names.
The measurement itself lives in bench.py (run_c5) so that the driver times the same code."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--aggregates", type=int, default=10_000_000)
    ap.add_argument("--batch-events", type=int, default=100_000)
    ap.add_argument("--batches", type=int, default=600)
    ap.add_argument("--snapshot-every", type=int, default=30, help="publish a state-topic delta every N batches (0 = never)")
    ap.add_argument("--device-batches", action="store_true", help="batches already in HBM (no staging / H2D)")
    a = ap.parse_args()
    ns = argparse.Namespace(aggregates=a.aggregates, batch_events=a.batch_events, steps=a.batches, warmup=3, snapshot_every=a.snapshot_every,
                            device_batches=a.device_batches, no_cpu_baseline=False, parity="full")
    print(json.dumps(bench.run_c5(ns)))
