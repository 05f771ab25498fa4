"""Multi-GPU layout: one process per GPU, aggregates sharded by the reference's own shard map.

The path shards naturally — aggregates are independent (one writer per aggregate,
``modules/surge-docs/src/main/paradox/overview.md:37-39``):

    partition = abs(MurmurHash3.stringHash(aggregateId.takeWhile(_ != ':')) % numPartitions)
        — modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8,38-42
    gpu       = partition % world_size          (the reference assigns partitions to nodes through the
                                                 Kafka Streams consumer group, PartitionAssignments.scala:51-63)

Each rank folds its shard with no communication.  The single exchange step is the all-gather of the
final snapshot (``A_r x 64`` bytes per rank, counts differ per rank, so shards are padded to the
max count) over RCCL/xGMI — ``torch.distributed`` with backend ``nccl`` on GPUs, ``gloo`` in the CPU
tests.  The gather runs on a side stream so it overlaps the next replay's fold.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

ID_PREFIX = "acct-"
ID_DIGITS = 8


def aggregate_id(i: int) -> str:
    """Synthetic aggregate ids of the sharded configs: ``acct-%08d``."""
    return f"{ID_PREFIX}{i:0{ID_DIGITS}d}"


def shard_of_partition(partition, world_size: int):
    return partition % world_size


def id_table_utf16(ids, xp_device=None):
    """UTF-16 code units of ``acct-%08d`` for integer ids (numpy array or torch tensor) + offsets."""
    width = len(ID_PREFIX) + ID_DIGITS
    if type(ids).__module__.startswith("torch"):
        import torch

        n = ids.numel()
        out = torch.empty((n, width), dtype=torch.int16, device=ids.device)
        for k, ch in enumerate(ID_PREFIX):
            out[:, k] = ord(ch)
        for k in range(ID_DIGITS):
            out[:, len(ID_PREFIX) + k] = ((ids // (10 ** (ID_DIGITS - 1 - k))) % 10 + 48).to(torch.int16)
        off = torch.arange(n + 1, dtype=torch.int64, device=ids.device) * width
        return out.reshape(-1), off
    ids = np.asarray(ids, dtype=np.int64)
    n = ids.shape[0]
    out = np.empty((n, width), dtype=np.uint16)
    for k, ch in enumerate(ID_PREFIX):
        out[:, k] = ord(ch)
    for k in range(ID_DIGITS):
        out[:, len(ID_PREFIX) + k] = (ids // (10 ** (ID_DIGITS - 1 - k))) % 10 + 48
    return out.reshape(-1), np.arange(n + 1, dtype=np.int64) * width


def partitions_of_ids(ids, n_partitions: int, engine=None):
    """``partitionForKey`` of ``acct-%08d`` ids.  CUDA tensors go through kernel K4 (needs ``engine``),
    host arrays through the C ABI's CPU entry point."""
    if type(ids).__module__.startswith("torch") and ids.is_cuda:
        import torch

        if engine is None:
            raise ValueError("device ids need a ReplayEngine to launch the hash kernel on")
        utf16, off = id_table_utf16(ids)
        out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
        torch.cuda.current_stream(ids.device).synchronize()  # the id table was built on torch's stream, the kernel runs on the engine's
        engine.partition_hash_device(utf16, off, n_partitions, out)
        engine.synchronize()
        return out
    import ctypes

    from . import _native

    host = ids.numpy() if type(ids).__module__.startswith("torch") else np.asarray(ids, dtype=np.int64)
    utf16, off = id_table_utf16(host)
    out = np.zeros(host.shape[0], dtype=np.int32)
    if host.shape[0]:
        rc = _native.load().surge_replay_partition_hash(
            utf16.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), host.shape[0], n_partitions,
            out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"surge_replay_partition_hash failed: {rc}")
    return out


def local_aggregate_ids(n_global: int, n_partitions: int, rank: int, world_size: int, device="cpu", engine=None):
    """Global indices (ascending) of the aggregates rank ``rank`` owns."""
    if str(device).startswith("cuda"):
        import torch

        ids = torch.arange(n_global, dtype=torch.int64, device=device)
        part = partitions_of_ids(ids, n_partitions, engine).to(torch.int64)
        return ids[shard_of_partition(part, world_size) == rank]
    ids = np.arange(n_global, dtype=np.int64)
    part = partitions_of_ids(ids, n_partitions).astype(np.int64)
    return ids[shard_of_partition(part, world_size) == rank]


class NativeSnapshotGather:
    """The exchange through the C ABI (``surge_replay_comm_*`` / ``surge_replay_allgather_snapshot``): RCCL inside
    ``libsurge_replay.so``, no ``torch.distributed`` on the data path.  ``torch.distributed`` (any backend) is only the
    out-of-band channel that hands rank 0's 128-byte communicator id to the other ranks — a JVM host would use
    whatever channel it has (the Kafka Streams group metadata, a config topic, ...).

    Same surface as :class:`SnapshotGather`: two output slots alternate so a gather can still be in flight on the
    library's side stream while the next fold writes the other snapshot buffer."""

    def __init__(self, n_local: int, device, engine, group=None, mode: Optional[str] = None):
        import os

        import torch
        import torch.distributed as dist

        self.torch, self.engine = torch, engine
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # p2p_raw (default): per-peer send / recv of the 64-byte states, no pack / expand passes — least HBM traffic, which is
        # what an exchange overlapped with the HBM-bound fold competes for; p2p / allgather: the 40-byte wire form
        self.mode = mode or os.environ.get("SURGE_SNAPSHOT_GATHER", "p2p_raw")
        if self.mode not in ("p2p", "p2p_raw", "allgather"):
            raise ValueError(f"unknown snapshot gather mode {self.mode!r}")
        self.packed = self.mode != "p2p_raw"
        uid = [engine.comm_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(uid, src=0, group=group)
        engine.comm_init(self.rank, self.world, uid[0])
        counts, self.max_count = engine.comm_counts(n_local)
        self.counts = [int(c) for c in counts]
        self.n_local = n_local
        self.out = [torch.zeros((self.world, self.max_count, 64), dtype=torch.uint8, device=self.device) for _ in range(2)]

    def make_local_buffers(self):
        return [self.torch.zeros((self.max_count, 64), dtype=self.torch.uint8, device=self.device) for _ in range(2)]

    def launch(self, slot: int, local_padded, ready_event=None) -> None:
        # the library orders the exchange after everything already enqueued on the engine's fold stream
        self.engine.allgather_snapshot(local_padded, self.n_local, self.out[slot], self.max_count, slot,
                                       {"p2p": 0, "allgather": 1, "p2p_raw": 2}[self.mode])

    def wait(self, slot: int, stream=None) -> None:
        self.engine.comm_wait(slot)  # the engine's fold stream waits for that slot's exchange

    def synchronize(self, slot: int) -> None:
        self.engine.comm_wait(slot, host_sync=True)

    def result(self, slot: int):
        return self.out[slot]

    def info(self) -> dict:
        return self.engine.comm_info()


class SnapshotGather:
    """All-gather of the per-rank final snapshots, overlapped with the next fold.

    ``launch(local_states, ready_event)`` enqueues the collective on a side stream after
    ``ready_event``; ``result(slot)`` is ``[world, max_count, 64]`` (rows beyond a rank's count are
    padding).  Two output slots alternate so a gather can still be in flight while the next fold
    writes the other snapshot buffer.
    """

    def __init__(self, n_local: int, device, group=None, mode: Optional[str] = None, engine=None, packed: Optional[bool] = None):
        import os

        import torch
        import torch.distributed as dist

        # "p2p": one grouped send/recv per peer — on the xGMI full mesh every one of the 7 links then carries
        # exactly one peer's shard concurrently (a ring all-gather would push all N-1 shards through one link
        # pair).  "allgather": the library collective.
        self.mode = mode or os.environ.get("SURGE_SNAPSHOT_GATHER", "p2p")
        # shards travel in the 40-byte wire form (the 24-byte reserved tail of a state is always zero) and are
        # expanded back to 64 bytes on arrival: 37.5 % less xGMI traffic; "p2p_raw" = p2p without that (64-byte states)
        self.engine = engine
        self.packed = (os.environ.get("SURGE_SNAPSHOT_PACKED", "1") == "1") if packed is None else packed
        if self.mode == "p2p_raw":
            self.mode, self.packed = "p2p", False

        self.dist, self.torch = dist, torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        counts = torch.tensor([n_local], dtype=torch.int64, device=self.device)
        all_counts = [torch.zeros_like(counts) for _ in range(self.world)]
        dist.all_gather(all_counts, counts, group=group)
        self.counts = [int(c.item()) for c in all_counts]
        self.max_count = max(self.counts)
        self.n_local = n_local
        self.out = [torch.zeros((self.world, self.max_count, 64), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.engine is None:
            self.packed = False  # the pack / unpack kernels are launched through the engine's C ABI
        if self.packed:
            self.wire = [torch.zeros((self.world, self.max_count, 40), dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.local_wire = [torch.zeros((self.max_count, 40), dtype=torch.uint8, device=self.device) for _ in range(2)]
        # gloo cannot all-gather device tensors: stage through the host (debug / single-GPU rehearsal only)
        self.stage_host = self.cuda and dist.get_backend(group) == "gloo"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.done = [None, None]
        if self.cuda and not self.stage_host and self.mode == "p2p" and self.world > 1:
            self._probe_p2p()

    def _probe_p2p(self) -> None:
        """One tiny grouped send/recv round before the first real exchange: if this RCCL build rejects it on
        any rank, every rank switches to the library all-gather (the decision is agreed by an all-reduce)."""
        import sys

        torch, dist = self.torch, self.dist
        ok = 1
        try:
            dst = torch.zeros((self.world, 16), dtype=torch.uint8, device=self.device)
            src = torch.full((16,), self.rank, dtype=torch.uint8, device=self.device)
            self._exchange(dst, src)
            torch.cuda.synchronize(self.device)
            if not bool((dst == torch.arange(self.world, dtype=torch.uint8, device=self.device)[:, None]).all()):
                ok = 0
        except Exception as exc:  # pragma: no cover - depends on the RCCL build
            print(f"[surge_amd.dist] grouped send/recv probe failed on rank {self.rank}: {exc}", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            self.mode = "allgather"

    def make_local_buffers(self):
        """Two snapshot buffers padded to ``max_count`` rows (the fold writes the first ``n_local``)."""
        return [self.torch.zeros((self.max_count, 64), dtype=self.torch.uint8, device=self.device) for _ in range(2)]

    def launch(self, slot: int, local_padded, ready_event=None) -> None:
        if self.stage_host:
            if ready_event is not None:
                ready_event.synchronize()
            host = local_padded.cpu()
            parts = [self.torch.zeros_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host, group=self.group)
            self.out[slot].copy_(self.torch.stack(parts))
        elif self.cuda:
            with self.torch.cuda.stream(self.stream):
                if ready_event is not None:
                    self.stream.wait_event(ready_event)
                if self.packed:
                    self.engine.pack_states(local_padded, self.local_wire[slot], stream=self.stream)
                    self._exchange(self.wire[slot], self.local_wire[slot])
                    self.engine.unpack_states(self.wire[slot].view(-1, 40), self.out[slot].view(-1, 64), stream=self.stream)
                else:
                    self._exchange(self.out[slot], local_padded)
                ev = self.torch.cuda.Event()
                ev.record(self.stream)
                self.done[slot] = ev
        elif self.packed:
            self.local_wire[slot].copy_(local_padded[:, :40])
            self._exchange(self.wire[slot], self.local_wire[slot])
            self.out[slot][:, :, :40] = self.wire[slot]
            self.out[slot][:, :, 40:] = 0
        else:
            self._exchange(self.out[slot], local_padded)

    def _exchange(self, dst, src) -> None:
        """dst[r] := rank r's src, for every rank."""
        dist = self.dist
        if self.mode != "p2p":
            if self.cuda:
                dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=self.group)
            else:
                dist.all_gather([dst[r] for r in range(self.world)], src, group=self.group)
            return
        dst[self.rank].copy_(src)
        ops = []
        for d in range(1, self.world):  # skewed peer order: rank r talks to r+d / r-d in step d
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            ops.append(dist.P2POp(dist.isend, src, to, self.group))
            ops.append(dist.P2POp(dist.irecv, dst[frm], frm, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def wait(self, slot: int, stream=None) -> None:
        """Make ``stream`` (default: current) wait for slot ``slot``'s gather."""
        if self.cuda and self.done[slot] is not None:
            (stream or self.torch.cuda.current_stream(self.device)).wait_event(self.done[slot])

    def result(self, slot: int):
        return self.out[slot]

    def assemble(self, slot: int, owner_ids: List) -> "np.ndarray":
        """Host-side reassembly into global aggregate order (for checks): ``owner_ids[r]`` are the
        global indices rank r owns."""
        flat = self.out[slot].cpu().numpy()
        n_global = sum(len(o) for o in owner_ids)
        res = np.zeros((n_global, 64), dtype=np.uint8)
        for r, ids in enumerate(owner_ids):
            ids = np.asarray(ids.cpu() if hasattr(ids, "cpu") else ids)
            res[ids] = flat[r, : len(ids)]
        return res
