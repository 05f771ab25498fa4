"""N > 1 path on CPU: world_size 2, gloo.  Shard map, per-rank shard generation, max-padded all-gather
of the final snapshot and reassembly — everything bench.py does for --gpus N except the GPU fold
itself, which is replaced here by the CPU oracle (this is a test).  Second test: the INGEST sharded by partition (what
--workload e2e --gpus N does in front of the fold) on a reference-shaped transactional topic, through the library's host
framer / decoder."""
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


# ---- the ingest sharded by partition (what bench.py --workload e2e --gpus N and a restore consumer per GPU do) ------------------
E2E_AGGS, E2E_CAP, E2E_PART = 1500, 5, 16


def _e2e_events(agg, j):
    """the (aggregate, j) -> event function both the topic and the expectation are written from"""
    h = (agg.astype(np.int64) * 2654435761 + j.astype(np.int64) * 40503) & 0xFFFFFFFF
    return (h % 3).astype(np.int32), (h // 7 % 1000).astype(np.int32)


def _e2e_topic():
    """A transactional, reference-shaped topic of Counter events over E2E_PART partitions (independent writer): two fetch
    responses per partition list; returns (fetches, per-aggregate event counts)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import topic_gen
    from surge_amd.kafka import partition_for_keys

    ids = np.arange(E2E_AGGS, dtype=np.int64)
    counts = 1 + (ids * 7919 % E2E_CAP)
    part = np.asarray(partition_for_keys([f"acct-{i:08d}" for i in ids], E2E_PART), dtype=np.int32)
    fetches = []
    with topic_gen.WireTopic(E2E_PART, flush_events=40, max_batch_bytes=2048, codec="lz4", abort_every=5, hold_markers=3) as topic:
        for j in range(1, E2E_CAP + 1):
            sel = ids[counts >= j]
            ty, arg = _e2e_events(sel, np.full(sel.shape[0], j))
            k, ko, v, vo = topic_gen.counter_records(sel, ty, arg, np.full(sel.shape[0], j, np.int32))
            fetches.append(topic.fetch(part[sel], k, ko, v, vo, last=j == E2E_CAP))
    return fetches, counts


def _e2e_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fixture_models import CounterBusinessLogic
    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd.ingest import EventsTopicIngest

    model = CounterBusinessLogic().command_model()
    tmpl = model.event_json_template()
    fetches, _ = _e2e_topic()
    mine = [p for p in range(E2E_PART) if p % world == rank]  # PartitionAssignments.scala:51-63 as bench.py shards them
    keys, aggs, evs = [], [], []
    for p in mine:  # one read_committed framer + host decoder per partition; ids interned per partition, re-keyed below
        with EventsTopicIngest() as g:
            for f in fetches:
                if f[p]:
                    g.feed(f[p])
            a, e, _ = g.drain_json(tmpl)
            k = g.key_table().keys
        base = len(keys)
        keys += k
        aggs.append(a + base)
        evs.append(e)
    agg = np.concatenate(aggs) if aggs else np.zeros(0, np.int64)
    ev = np.concatenate(evs) if evs else np.zeros(0, S.EVENT_DTYPE)
    order = np.argsort(agg, kind="stable")  # offset order inside a partition = event order of its aggregates
    off = np.zeros(len(keys) + 1, np.int64)
    np.cumsum(np.bincount(agg, minlength=len(keys)), out=off[1:])
    states = oracle.fold_csr(off, ev[order], None, model.event_algebra())
    n = torch.tensor([len(keys), int(ev.shape[0])], dtype=torch.int64)
    both = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(both, n)
    np.save(os.path.join(out_dir, f"e2e_states{rank}.npy"), states.view(np.uint8).reshape(-1, 64))
    with open(os.path.join(out_dir, f"e2e_keys{rank}.txt"), "w") as fh:
        fh.write("\n".join(keys))
    np.save(os.path.join(out_dir, f"e2e_totals{rank}.npy"), torch.stack(both).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_ingest_sharded_by_partition_recovers_every_aggregate_once(tmp_path):
    """bench.py --workload e2e --gpus N on CPU plumbing: rank r frames and decodes the partitions p % N == r of a topic shaped like
    the reference's (transactions, aborted flushes, late markers; the independent writer), folds what it decoded (the oracle
    stands in for the GPU fold: this is a test), and the ranks' key tables partition the aggregates; every state equals the fold
    of the SOURCE events."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from fixture_models import CT_DEC, CT_INC, CT_NOOP, CounterBusinessLogic
    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd.kafka import partition_for_keys

    world, port = 2, _free_port()
    mp.start_processes(_e2e_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    _, counts = _e2e_topic()
    model = CounterBusinessLogic().command_model()
    seen = {}
    for r in range(world):
        keys = open(tmp_path / f"e2e_keys{r}.txt").read().split("\n")
        states = np.load(tmp_path / f"e2e_states{r}.npy")
        totals = np.load(tmp_path / f"e2e_totals{r}.npy")
        assert totals[:, 0].sum() == E2E_AGGS and totals[:, 1].sum() == counts.sum()  # both ranks agree on the job's totals
        assert len(keys) == states.shape[0] == totals[r, 0]
        parts = partition_for_keys(keys, E2E_PART)
        assert all(p % world == r for p in parts)
        for k, st in zip(keys, states):
            assert k not in seen
            seen[k] = st.tobytes()
    assert sorted(seen) == [f"acct-{i:08d}" for i in range(E2E_AGGS)]
    # the expectation from the source events, aggregate by aggregate
    ids = np.arange(E2E_AGGS, dtype=np.int64)
    off = np.zeros(E2E_AGGS + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    agg = np.repeat(ids, counts)
    j = (np.arange(off[-1]) - np.repeat(off[:-1], counts) + 1).astype(np.int32)
    ty, arg = _e2e_events(agg, j)
    src = np.zeros(off[-1], dtype=S.EVENT_DTYPE)
    src["type"] = np.array([CT_INC, CT_DEC, CT_NOOP], np.int32)[ty]
    src["seq"] = j
    src["raw"] = arg.astype(np.uint32).astype(np.uint64)
    exp = oracle.fold_csr(off, src, None, model.event_algebra()).view(np.uint8).reshape(-1, 64)
    for i in range(E2E_AGGS):
        assert seen[f"acct-{i:08d}"] == exp[i].tobytes(), i
