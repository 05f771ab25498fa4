#!/usr/bin/env python3
"""FLAT on small / short-rowed logs: task granularity (SURGE_REPLAY_TARGET_TASKS, SURGE_REPLAY_TASK_KB) and tile size
(SURGE_REPLAY_LE_FLAT) against the defaults; median of 20 folds after 3 warm-ups.  (needs a GPU)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from surge_amd import schema as S
from surge_amd import synth
from surge_amd.replay import ReplayEngine

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(11)
KNOBS = ("SURGE_REPLAY_TARGET_TASKS", "SURGE_REPLAY_TASK_KB", "SURGE_REPLAY_LE_FLAT")
SETTINGS = [{}, {"SURGE_REPLAY_LE_FLAT": "8"}]
for tt in ("2048", "4096", "8192", "32768", "65536"):
    SETTINGS.append({"SURGE_REPLAY_TARGET_TASKS": tt})
    SETTINGS.append({"SURGE_REPLAY_TARGET_TASKS": tt, "SURGE_REPLAY_LE_FLAT": "8"})
for kind, n in (("zipf", 100_000), ("zipf", 150_000), ("zipf", 300_000), ("uniform_1_32", 2_000_000), ("uniform_1_32", 20_000_000)):
    if kind == "zipf":
        lens = synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3).to(torch.int64)
    else:
        lens = torch.randint(1, 33, (n,), device=dev, generator=g).to(torch.int64)
    so, ev = synth.csr_log_device(lens, 3)
    row = {"shape": kind, "aggregates": n, "events": int(so[-1])}
    with ReplayEngine() as eng:
        eng.load_csr(so, ev)
        for st_ in SETTINGS:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(st_)
            for _ in range(3):
                eng.fold(S.ALGO_FLAT)
            eng.synchronize()
            eng.stats_reset()
            for _ in range(20):
                eng.fold(S.ALGO_FLAT)
            eng.synchronize()
            st = eng.stats()
            t = eng.fold_times_ms()
            ms = float(np.median(t))
            row[",".join(f"{k.split('_')[-1]}={v}" for k, v in st_.items()) or "default"] = {
                "ms": round(ms, 4), "min_ms": round(float(np.min(t)), 4), "frac": round(st.algorithmic_bytes / (ms * 1e-3) / 8e12, 4), "tasks": int(st.n_tasks)}
    for k in KNOBS:
        os.environ.pop(k, None)
    del so, ev
    torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
