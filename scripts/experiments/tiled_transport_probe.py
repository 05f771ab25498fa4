"""TILED / SORTED / ROWS with and without the walk's arithmetic (libsurge_replay_exp2.so: -DSURGE_EXP_SKIP_APPLY_AOT), waves per CU swept."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from surge_amd import schema as S, synth
from surge_amd.replay import ReplayEngine
dev = torch.device("cuda:0")
n = 10_000_000
so, ev = synth.csr_log_device(synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), 3), 3)
out = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
with ReplayEngine() as e:
    e.load_csr(so, ev, None, out)
    for algo, name in ((S.ALGO_TILED, "tiled"), (S.ALGO_SORTED, "sorted")):
        for waves in ("", "3", "4", "5", "6", "8"):
            for subs in (("2", "1") if algo == S.ALGO_TILED else ("",)):
                os.environ.pop("SURGE_REPLAY_TILED_WAVES", None); os.environ.pop("SURGE_REPLAY_SORTED_WAVES", None)
                if waves: os.environ["SURGE_REPLAY_TILED_WAVES" if algo == S.ALGO_TILED else "SURGE_REPLAY_SORTED_WAVES"] = waves
                if subs: os.environ["SURGE_REPLAY_TILED_SUBS"] = subs
                e.fold(algo); e.synchronize(); e.stats_reset()
                for _ in range(4): e.fold(algo)
                e.synchronize()
                t = float(np.median(e.fold_times_ms()))
                print(json.dumps({"lib": os.path.basename(os.environ.get("SURGE_REPLAY_LIB", "main")), "algo": name, "waves_per_cu": waves or "default", "subs": subs, "ms": t, "frac": e.stats().algorithmic_bytes / t / 8e9}), flush=True)
