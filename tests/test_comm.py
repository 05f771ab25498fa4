"""The snapshot exchange behind the C ABI (SURVEY §8b surge_replay_allgather / §8e): RCCL inside libsurge_replay.so,
driven WITHOUT torch.distributed — what a JVM host would do through JNI."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "comm_worker.py")


def run_ranks(world, tmp_path, devices, mode=0, timeout=240):
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(tmp_path), str(devices[r]), str(mode)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append((p.returncode, out))
    except subprocess.TimeoutExpired:
        for p in procs:  # exactly the processes started here
            p.kill()
        outs = [(p.wait(), "timeout") for p in procs]
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_c_abi_exchange_single_rank(tmp_path, mode):
    (rc, out), = run_ranks(1, tmp_path, [0], mode)
    assert rc == 0 and "OK 0" in out, out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_c_abi_exchange_two_ranks(tmp_path, mode):
    """Two communicator ranks, one process each.  With two GPUs visible they take one each; on a one-GPU box both sit
    on cuda:0, which RCCL may refuse ("duplicate GPU") — then the test is skipped, loudly, with RCCL's own message."""
    import torch

    n = torch.cuda.device_count()
    if n < 2 and os.environ.get("SURGE_TEST_FORCE_2RANK") != "1":
        # measured on the 1-GPU MI355X boxes (RCCL 2.26.6): ncclCommInitRank answers "invalid usage" for the second rank on
        # the same device and leaves rank 0 waiting in the bootstrap until it is killed (~50 s per attempt): do not burn that
        pytest.skip("one GPU visible: RCCL refuses two communicator ranks on one device (ncclCommInitRank: invalid usage); "
                    "SURGE_TEST_FORCE_2RANK=1 tries anyway")
    outs = run_ranks(2, tmp_path, [0, 1] if n >= 2 else [0, 0], mode, timeout=120)
    if any(rc == 3 for rc, _ in outs) or (n < 2 and any(rc != 0 for rc, _ in outs)):
        pytest.skip("RCCL refused two ranks on this box's single GPU: " + " | ".join(o.strip().splitlines()[-1] for _, o in outs if o.strip()))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"OK {r}" in out, out
