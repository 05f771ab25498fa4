"""The snapshot exchange behind the C ABI (SURVEY §8b surge_replay_allgather / §8e): RCCL inside libsurge_replay.so,
driven WITHOUT torch.distributed — what a JVM host would do through JNI."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "comm_worker.py")


def run_ranks(world, tmp_path, devices, mode=0, timeout=240, env=None):
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(tmp_path), str(devices[r]), str(mode)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
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
@pytest.mark.parametrize("mode", [0, 1, 2])
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


def build_rccl_stub():
    """tests/rccl_stub/librccl_stub.so: the ten nccl* symbols comm.hip binds, with a file transport (two ranks, one GPU)."""
    d = os.path.join(HERE, "rccl_stub")
    src, lib = os.path.join(d, "rccl_stub.cpp"), os.path.join(d, "librccl_stub.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wl,-Bsymbolic", src, "-o", lib],
                       check=True, capture_output=True)
    return lib


def test_rccl_stub_builds_and_exports_what_the_product_binds():
    import ctypes
    import re

    lib = ctypes.CDLL(build_rccl_stub(), mode=os.RTLD_LOCAL)
    src = open(os.path.join(os.path.dirname(HERE), "surge_amd", "csrc", "comm.hip")).read()
    bound = set(re.findall(r'sym\(lib, "(nccl\w+)"', src))
    assert len(bound) == 10
    for name in bound:
        assert hasattr(lib, name), f"the stub lacks {name}"


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, 0), (2, 1), (2, 2), (3, 0), (3, 2)])  # mode 2 = 64-byte states, no pack / expand
def test_c_abi_exchange_several_ranks_one_gpu_over_the_stub_transport(tmp_path, world, mode):
    """The N > 1 code of comm.hip with real processes: SURGE_RCCL_LIBRARY points the product at tests/rccl_stub, whose
    transport (files) does not mind that every rank sits on cuda:0.  Counts exchange, per-peer send/recv pairing, ragged
    shards, None padding, both slots and a changed shard size all run as they would over RCCL."""
    env = dict(os.environ, SURGE_RCCL_LIBRARY=build_rccl_stub(), SURGE_RCCL_STUB_DIR=str(tmp_path))
    outs = run_ranks(world, tmp_path, [0] * world, mode, timeout=240, env=env)
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"OK {r} rccl=1 " in out and "librccl_stub.so" in out, out


# ---- in-process group: one host process drives every rank (surge_replay_allgather, the literal SURVEY §8b form) ----
def _group_shard(rank, world):
    import numpy as np

    from surge_amd import synth

    ids = np.arange(4000, dtype=np.int64)
    mine = ids[ids % world == rank][: 500 + 61 * rank]  # ragged shard sizes: all-gather-v
    so, ev = synth.csr_log(synth.zipf_lengths(mine, 5, max_len=300), 200 + rank, synth.STRESS_MIX)
    return so, ev


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_in_process_group_gathers_every_shard_on_every_handle(world):
    """All ranks in ONE process (handles on whatever GPUs are visible — all on cuda:0 on a one-GPU box): peer copies
    instead of RCCL, same result contract as the RCCL exchange."""
    import numpy as np
    import torch

    from oracle import oracle
    from surge_amd.replay import ReplayEngine, ReplayError

    ndev = torch.cuda.device_count()
    engines = [ReplayEngine(device=r % ndev) for r in range(world)]
    try:
        shards = [_group_shard(r, world) for r in range(world)]
        exp = [oracle.fold_csr(so, ev).view(np.uint8).reshape(-1, 64) for so, ev in shards]
        for e, (so, ev) in zip(engines, shards):
            e.load_csr(so, ev)
            e.fold()
        mx = max(x.shape[0] for x in exp)
        for slot in (0, 1, 0):  # handle-owned result, both slots, a slot reused
            ReplayEngine.allgather_group(engines, slot=slot)
            for d, e in enumerate(engines):
                info = e.comm_info()
                assert info["rank"] == d and info["world"] == world and info["rccl_version"] == 0 and "in-process" in info["library"]
                counts, m = e.comm_counts(0)
                assert list(counts) == [x.shape[0] for x in exp] and m == mx
                for r in range(world):
                    got = e.gathered_read(slot, r, 0, mx).view(np.uint8).reshape(-1, 64)
                    assert got[: exp[r].shape[0]].tobytes() == exp[r].tobytes(), f"handle {d}: shard of rank {r} differs"
                    assert not got[exp[r].shape[0]:].any(), "padding rows must be None (zero)"
        # caller-owned outputs, wider than needed, and only a prefix of every shard
        part = [x.shape[0] // 2 for x in exp]
        rows = max(part) + 3
        outs = [torch.full((world, rows, 64), 0xAB, dtype=torch.uint8, device=f"cuda:{r % ndev}") for r in range(world)]
        ReplayEngine.allgather_group(engines, n_local=part, outs=outs, rows_per_rank=rows, slot=1)
        for d, e in enumerate(engines):
            e.comm_wait(1, host_sync=True)
            got = outs[d].cpu().numpy()
            for r in range(world):
                assert got[r, : part[r]].tobytes() == exp[r][: part[r]].tobytes()
                assert not got[r, part[r]: max(part)].any(), "rows between a shard and the largest shard are None"
                assert (got[r, max(part):] == 0xAB).all(), "rows beyond the largest shard are not touched"
        # the next fold overlaps the exchange: fold new events, exchange again, the snapshot moves on
        for e, (so, ev) in zip(engines, shards):
            e.load_csr(so, ev, e.snapshot())
            e.fold()
        ReplayEngine.allgather_group(engines, slot=0)
        for r, (so, ev) in enumerate(shards):
            exp2 = oracle.fold_csr(so, ev, oracle.fold_csr(so, ev))
            got = engines[(r + 1) % world].gathered_read(0, r, 0, exp[r].shape[0])
            assert got.tobytes() == exp2.tobytes()
        # error paths: duplicate handle; a handle that holds an RCCL rank is refused
        if world > 1:
            with pytest.raises(ReplayError):
                ReplayEngine.allgather_group([engines[0], engines[0]])
    finally:
        for e in engines:
            e.close()


@pytest.mark.gpu
def test_in_process_group_moves_v2_states_whole():
    """Slot schemas use all 64 bytes of a state: no 40-byte wire form."""
    import numpy as np

    from oracle import oracle
    from surge_amd.replay import ReplayEngine
    from test_slots import LEDGER, LG_CLOSE, LG_CREDIT, LG_DEBIT, LG_OPEN, make_log

    rng = np.random.default_rng(11)
    types, p = [LG_OPEN, LG_CREDIT, LG_DEBIT, LG_CLOSE], [0.03, 0.55, 0.415, 0.005]
    engines = [ReplayEngine(LEDGER) for _ in range(2)]
    try:
        exp = []
        for e, n_agg in zip(engines, (700, 333)):
            so, ev = make_log(rng, n_agg, 60, types, p, (LG_OPEN, LG_CREDIT, LG_DEBIT))
            e.load_csr(so, ev)
            e.fold()
            exp.append(oracle.fold_csr_v2(so, ev, LEDGER))
        ReplayEngine.allgather_group(engines)
        for e in engines:
            for r in range(2):
                got = e.gathered_read(0, r, 0, 700)
                assert got[: exp[r].shape[0]].tobytes() == exp[r].tobytes()
                assert not got[exp[r].shape[0]:].view(np.uint8).any()
        with ReplayEngine() as v1:
            so, ev = _group_shard(0, 1)
            v1.load_csr(so, ev)
            v1.fold()
            with pytest.raises(Exception):
                ReplayEngine.allgather_group([engines[0], v1])
    finally:
        for e in engines:
            e.close()
