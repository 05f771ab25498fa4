import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surge_amd import synth
from surge_amd.replay import ReplayEngine
dev = torch.device("cuda:0")
A, L = int(os.environ.get("AGGS", "1000000")), int(os.environ.get("LEN", "256"))
so, ev = synth.fixed_log_device(A, L, 2, dev)
out = torch.empty((A, 64), dtype=torch.uint8, device=dev)
eng = ReplayEngine()
eng.load_csr(so, ev, None, out)
for algo in [int(a) for a in os.environ.get("ALGOS", "3,1,2").split(",")]:
    for _ in range(3): eng.fold(algo)
    eng.synchronize(); eng.stats_reset()
    for _ in range(10): eng.fold(algo)
    st = eng.stats()
    ms = st.sum_fold_kernel_ms / st.timed_folds
    print(f"{os.environ.get('SURGE_REPLAY_LIB','default')[-20:]} A={A} L={L} algo={algo}->{st.last_algo}: kernel {ms:.3f} ms {st.algorithmic_bytes/ms/1e6:.0f} GB/s")
