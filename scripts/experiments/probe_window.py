"""The bare read stream (surge_replay_stream_probe: fastest of six variants) over windows of 1 .. 74 GB of one 74 GB buffer."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from surge_amd.replay import ReplayEngine
dev = torch.device("cuda:0")
n = 4_604_728_291
buf = torch.empty((n, 2), dtype=torch.int64, device=dev)
buf.random_(0, 1 << 40)
with ReplayEngine() as e:
    for gb in (1, 4, 8, 16, 32, 48, 64, 73.6):
        vec = int(gb * 1e9 / 16)
        for off in ((0,) if gb > 40 else (0, (n - vec) // 2)):
            ms = min(e.stream_probe_ms(buf[off:off + vec]) for _ in range(3))
            print(json.dumps({"window_GB": gb, "offset_vec": off, "ms": ms, "GBps": vec * 16 / ms / 1e6, "frac_of_8TBps": vec * 16 / ms / 1e6 / 8000}), flush=True)
