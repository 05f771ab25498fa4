"""N > 1 path on CPU: world_size 2, gloo.  Shard map, per-rank shard generation, max-padded all-gather
of the final snapshot and reassembly — everything bench.py does for --gpus N except the GPU fold
itself, which is replaced here by the CPU oracle (this is a test)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N_GLOBAL, L, N_PART, SEED = 3000, 32, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode, packed):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from surge_amd import synth
    from surge_amd.dist import SnapshotGather, local_aggregate_ids

    ids = torch.from_numpy(local_aggregate_ids(N_GLOBAL, N_PART, rank, world, "cpu"))
    seg_off, events = synth.fixed_log_for_aggregates_device(ids, L, SEED)
    states = oracle.fold_csr(seg_off.numpy(), synth.to_event_records(events))
    gather = SnapshotGather(int(ids.numel()), "cpu", mode=mode, packed=packed)
    bufs = gather.make_local_buffers()
    bufs[0][: ids.numel()] = torch.from_numpy(states.view(np.uint8).reshape(-1, 64))
    gather.launch(0, bufs[0])
    all_ids = [torch.zeros(gather.counts[r], dtype=torch.int64) for r in range(world)]
    padded = torch.zeros(gather.max_count, dtype=torch.int64)
    padded[: ids.numel()] = ids
    parts = [torch.zeros(gather.max_count, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, padded)
    owner_ids = [parts[r][: gather.counts[r]].numpy() for r in range(world)]
    snap = gather.assemble(0, owner_ids)
    np.save(os.path.join(out_dir, f"snap{rank}.npy"), snap)
    np.save(os.path.join(out_dir, f"ids{rank}.npy"), ids.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,packed", [("p2p", True), ("allgather", False), ("p2p", False)])
def test_two_rank_sharded_replay_and_all_gather(tmp_path, mode, packed):
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from surge_amd import synth
    from surge_amd.dist import aggregate_id, partitions_of_ids
    from surge_amd.kafka import partition_for_keys

    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), mode, packed), nprocs=world, join=True, start_method="spawn")
    so, ev = synth.fixed_log(N_GLOBAL, L, SEED)
    expected = oracle.fold_csr(so, ev).view(np.uint8).reshape(-1, 64)
    ids = [np.load(tmp_path / f"ids{r}.npy") for r in range(world)]
    # the shards partition the aggregates, by the reference's partitioner
    assert sorted(np.concatenate(ids).tolist()) == list(range(N_GLOBAL))
    parts = partition_for_keys([aggregate_id(i) for i in range(N_GLOBAL)], N_PART)
    assert (partitions_of_ids(np.arange(N_GLOBAL), N_PART) == parts).all()
    for r in range(world):
        assert ((parts[ids[r]] % world) == r).all()
        # every rank ends up with the full, identical, correct snapshot
        snap = np.load(tmp_path / f"snap{r}.npy")
        assert snap.tobytes() == expected.tobytes()
