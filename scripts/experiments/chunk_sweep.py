"""GPU sweep for the chunked-rows kernel: Zipf logs of several sizes, algos FLAT / SORTED / CHUNKED, chunk targets T and
tile widths.  One process per (size): the chunk table is built once per bound log, so T changes need a re-bind."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from surge_amd import synth
from surge_amd.replay import ReplayEngine

dev = torch.device("cuda:0")
sizes = [int(x) for x in os.environ.get("ZIPF_SIZES", "500000,1250000,2000000").split(",")]
variants = os.environ.get("VARIANTS", "2:0:0,4:0:0,5:128:16,5:256:16,5:512:16,5:256:8").split(",")
for n in sizes:
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    lens = synth.zipf_lengths(ids, 3)
    so, ev = synth.csr_log_device(lens, 3)
    E = ev.shape[0]
    out = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    ref = None
    for v in variants:
        algo, T, le, *rest = (int(x) for x in v.split(":"))
        if T:
            os.environ["SURGE_REPLAY_CHUNK_T"] = str(T)
        else:
            os.environ.pop("SURGE_REPLAY_CHUNK_T", None)  # the engine's own choice
        if le:
            os.environ["SURGE_REPLAY_LE_CHUNKED"] = str(le)
        eng = ReplayEngine()
        eng.load_csr(so, ev, None, out)
        for _ in range(2):
            eng.fold(algo)
        eng.synchronize()
        eng.stats_reset()
        for _ in range(8):
            eng.fold(algo)
        st = eng.stats()
        ms = st.sum_fold_kernel_ms / st.timed_folds
        same = ""
        if ref is None:
            ref = out.clone()
        else:
            same = "same" if torch.equal(ref, out) else "DIFFERENT"
        print(f"zipf {n} aggs {E/1e6:.0f}M ev algo={algo} T={T} LE={le} waves={st.n_tasks}: {ms:.3f} ms "
              f"{st.algorithmic_bytes/ms/1e6:.0f} GB/s = {st.algorithmic_bytes/ms/1e6/80:.1f} % {same}", flush=True)
        eng.close()
    del so, ev, out, ref
    torch.cuda.empty_cache()
