"""Runtime-selectable variants (SURGE_DBG_TILED modes) on the SAME handles: H handles of the same library on the same bound
log (different allocations), rounds alternate over (handle, mode).  MODES=0,3,1 HANDLES=2 SHAPE=z4m SUBS WAVES ROUNDS FOLDS"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from surge_amd import synth
from surge_amd.replay import ReplayEngine

dev = torch.device("cuda:0")
modes = os.environ.get("MODES", "0,3,1").split(",")
H = int(os.environ.get("HANDLES", "2"))
shape = os.environ.get("SHAPE", "z4m")
algo = int(os.environ.get("ALGO", "7"))
rounds, folds = int(os.environ.get("ROUNDS", "6")), int(os.environ.get("FOLDS", "10"))
os.environ.setdefault("SURGE_REPLAY_TILED_SUBS", os.environ.get("SUBS", "2"))
os.environ.setdefault("SURGE_REPLAY_TILED_WAVES", os.environ.get("WAVES", "6"))
if shape == "c2":
    so, ev = synth.fixed_log_device(1_000_000, 256, 2, dev)
else:
    n = {"c4s": 1_250_000, "c3": 10_000_000, "z300k": 300_000, "z2m": 2_000_000, "z4m": 4_000_000, "z100k": 100_000}[shape]
    so, ev = synth.csr_log_device(synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3), 3)
n = so.numel() - 1
engines, outs = [], []
for h in range(H):
    e = ReplayEngine()
    outs.append(torch.zeros((n, 64), dtype=torch.uint8, device=dev))
    e.load_csr(so, ev, None, outs[h])
    e.fold(algo)
    e.synchronize()
    engines.append(e)
res = {}
for r in range(rounds):
    for h in range(H):
        for m in modes:
            os.environ["SURGE_DBG_TILED"] = m
            e = engines[h]
            e.stats_reset()
            for _ in range(folds):
                e.fold(algo)
            e.synchronize()
            res.setdefault((h, m), []).append(float(np.median(e.fold_times_ms())))
if os.environ.get("SURGE_DBG_PROBE_TILES"):
    for h in range(H):
        ms = min(engines[h].stream_probe_ms(ev) for _ in range(5))
        print(f"  handle {h}: stream probe over its tile-major copy {ms:.4f} ms")
ab = engines[0].stats().algorithmic_bytes
print(f"{shape}: {n} aggregates, algo {algo}, subs {os.environ['SURGE_REPLAY_TILED_SUBS']} waves/CU {os.environ['SURGE_REPLAY_TILED_WAVES']}, {rounds} rounds x {folds} folds")
for (h, m), x in sorted(res.items()):
    x = np.array(x)
    print(f"  handle {h} mode {m}: median {np.median(x):.4f} ms (min {x.min():.4f} max {x.max():.4f})  frac {ab / np.median(x) / 8e9:.4f}")
