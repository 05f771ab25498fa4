#!/usr/bin/env python3
"""Same-box A/B/A of the lane-per-row folds: ahead-of-time kernels (op table in LDS) against the kernels compiled for the
handle's op table (hiprtc, V1_LANES), built-in schema and the Counter fixture's own schema, 8- and 16-event lanes — every
variant its own handle on the same HBM-resident log, rounds of folds interleaved in time.

    SHAPE=c3|c4s|z2m|c2  ROUNDS=6 FOLDS=6  python scripts/lane_spec_ab.py  [> gpurun_out/lane_spec_ab.jsonl]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np
import torch

from fixture_models import COUNTER_ALGEBRA
from surge_amd import schema as S
from surge_amd import synth
from surge_amd.replay import ReplayEngine
from surge_amd.schema import DEFAULT_ALGEBRA

dev = torch.device("cuda:0")
shape = os.environ.get("SHAPE", "c3")
rounds, folds = int(os.environ.get("ROUNDS", "6")), int(os.environ.get("FOLDS", "6"))
schemas = os.environ.get("SCHEMAS", "builtin,counter").split(",")
les = os.environ.get("LES", "16,8").split(",")
algo_env = os.environ.get("ALGO")

logs = {}
for sc in schemas:
    mix = synth.C2_MIX if sc == "builtin" else synth.C1_MIX  # the Counter schema knows NOOP / INC / DEC only
    if shape == "c2":
        logs[sc] = synth.fixed_log_device(1_000_000, 256, 2, dev, mix)
    else:
        n = {"c4s": 1_250_000, "c3": 10_000_000, "z300k": 300_000, "z2m": 2_000_000, "z4m": 4_000_000}[shape]
        logs[sc] = synth.csr_log_device(synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3), 3, mix)
# VARIANTS="schema:build:le[:waves_per_cu[:lds_pad]],..." (default: every schema x aot / rtc x LES)
variants = []
if os.environ.get("VARIANTS"):
    for spec in os.environ["VARIANTS"].split(","):
        f = spec.split(":")
        variants.append((f[0], f[1], f[2], f[3] if len(f) > 3 else "", f[4] if len(f) > 4 else "", f[5] if len(f) > 5 else ""))
else:
    for sc in schemas:
        for build in ("aot", "rtc"):
            for le in les:
                variants.append((sc, build, le, "", "", ""))


def set_env(v):
    sc, build, le, waves, pad, extra = v
    os.environ["SURGE_REPLAY_RTC_LANES"] = "1" if build == "rtc" else "0"
    for name in ("SURGE_REPLAY_LE_SORTED", "SURGE_REPLAY_LE_CHUNKED", "SURGE_REPLAY_LE_ROWS"):
        os.environ[name] = le
    # (extra: #defines for the run-time compiled program, "A=1+B=1" — libsurge_replay_exp.so only, scripts/build_experiments_lib.py)
    for name, val in (("SURGE_REPLAY_SORTED_WAVES", waves), ("SURGE_REPLAY_RTC_LDS_PAD", pad), ("SURGE_REPLAY_RTC_EXTRA", extra.replace("+", ";"))):
        if val:
            os.environ[name] = val
        else:
            os.environ.pop(name, None)


engines, outs = {}, {}
for v in variants:
    set_env(v)
    sc = v[0]
    so, ev = logs[sc]
    e = ReplayEngine(DEFAULT_ALGEBRA if sc == "builtin" else COUNTER_ALGEBRA)
    outs[v] = torch.zeros((so.numel() - 1, 64), dtype=torch.uint8, device=dev)
    e.load_csr(so, ev, None, outs[v])
    algo = int(algo_env) if algo_env else S.ALGO_AUTO
    e.fold(algo)
    e.synchronize()
    engines[v] = (e, algo)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        set_env(v)  # (the knobs are read at every fold)
        e, algo = engines[v]
        e.stats_reset()
        for _ in range(folds):
            e.fold(algo)
        e.synchronize()
        res[v].append(float(np.median(e.fold_times_ms())))
first = {}
for v in variants:
    sc, build, le, waves, pad, extra = v
    e, algo = engines[v]
    st = e.stats()
    x = np.array(res[v])
    ref = outs[first.setdefault(sc, v)]
    print(json.dumps({"shape": shape, "schema": sc, "build": build, "lane_events": int(le), "waves_per_cu": waves or "default", "lds_pad": pad or 0, "extra": extra,
                      "algo": int(st.last_algo), "median_ms": float(np.median(x)),
                      "min_ms": float(x.min()), "max_ms": float(x.max()), "frac_of_8TBps": st.algorithmic_bytes / float(np.median(x)) / 8e9,
                      "rounds_ms": [round(float(t), 4) for t in x], "states_equal_first_variant": bool(torch.equal(outs[v], ref)),
                      "kernel_info": e.kernel_info()["detail"][:60]}), flush=True)
