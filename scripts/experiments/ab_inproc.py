"""A/B/C of library variants inside ONE process (same box, same allocations, interleaved in time): every variant gets its
own handle on the same bound log; rounds of FOLDS folds alternate between them.  VARIANTS=main,old  SHAPE=z4m  ALGO=7
SUBS=2 WAVES=6 ROUNDS=8 FOLDS=10"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from surge_amd import _native, synth
from surge_amd.replay import ReplayEngine

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
variants = os.environ.get("VARIANTS", "main,old").split(",")
shape = os.environ.get("SHAPE", "z4m")
algo = int(os.environ.get("ALGO", "7"))
rounds, folds = int(os.environ.get("ROUNDS", "8")), int(os.environ.get("FOLDS", "10"))
os.environ.setdefault("SURGE_REPLAY_TILED_SUBS", os.environ.get("SUBS", "2"))
os.environ.setdefault("SURGE_REPLAY_TILED_WAVES", os.environ.get("WAVES", "6"))
if shape == "c2":
    so, ev = synth.fixed_log_device(1_000_000, 256, 2, dev)
else:
    n = {"c4s": 1_250_000, "c3": 10_000_000, "z300k": 300_000, "z2m": 2_000_000, "z4m": 4_000_000, "z100k": 100_000}[shape]
    so, ev = synth.csr_log_device(synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3), 3)
n = so.numel() - 1
engines, outs = {}, {}
for v in variants:
    _native._lib = None
    os.environ["SURGE_REPLAY_LIB"] = os.path.join(root, "surge_amd", "libsurge_replay.so" if v == "main" else f"libsurge_replay_{v}.so")
    e = ReplayEngine()
    outs[v] = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    e.load_csr(so, ev, None, outs[v])
    e.fold(algo)
    e.synchronize()
    engines[v] = e
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        e = engines[v]
        e.stats_reset()
        for _ in range(folds):
            e.fold(algo)
        e.synchronize()
        res[v].append(float(np.median(e.fold_times_ms())))
ab = engines[variants[0]].stats().algorithmic_bytes
print(f"{shape}: {n} aggregates, algo {algo}, subs {os.environ['SURGE_REPLAY_TILED_SUBS']} waves/CU {os.environ['SURGE_REPLAY_TILED_WAVES']}, {rounds} rounds x {folds} folds")
chk0 = None
for v in variants:
    x = np.array(res[v])
    chk = int(outs[v].view(torch.int64).sum().item()) & 0xffffffff
    print(f"  {v:10s} median {np.median(x):.4f} ms (min {x.min():.4f} max {x.max():.4f})  frac {ab / np.median(x) / 8e9:.4f}  chk {chk:08x}   " + " ".join(f"{t:.3f}" for t in x))
