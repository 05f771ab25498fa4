"""``ReplayEngine`` — thin Python host over the C ABI (``include/surge_replay.h``).

What it replaces in the reference: Kafka Streams restoring the aggregate KTable
(``modules/common/src/main/scala/surge/kafka/streams/SurgeStateStoreConsumer.scala:57-76``)
followed by one ``readState`` per actor start; here the state store is rebuilt by folding the
events topic on the GPU —
``events.foldLeft(state)(handleEvent)`` (``.../scaladsl/command/CommandModels.scala:26``).

Host (numpy) buffers go through ``surge_replay_load_csr``; device buffers (torch tensors on
``cuda``) are bound zero-copy.  PyTorch is only plumbing here: device memory and streams.
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Optional

import numpy as np

from . import _native
from .schema import (
    ALGO_AUTO,
    DEFAULT_ALGEBRA,
    EVENT_DTYPE,
    STATE_DTYPE,
    CKernelInfo,
    CLayoutInfo,
    CSchema,
    CStats,
    EventAlgebra,
)

_STATUS = {0: "OK", -1: "INVALID", -2: "STATE", -3: "DEVICE", -4: "NOMEM", -5: "UNSUPPORTED", -6: "RANGE", -7: "CORRUPT", -8: "COMM"}


class ReplayError(RuntimeError):
    """A C-ABI call returned a negative status (maps to a failed ``Future`` on the JVM side)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"surge_replay status {status} ({_STATUS.get(status, '?')}): {message}")
        self.status = status


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _dev_ptr(t, nbytes_min: int = 0):
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("device buffers must be contiguous CUDA/HIP tensors")
    if t.numel() * t.element_size() < nbytes_min:
        raise ValueError("device buffer too small")
    return ctypes.c_void_p(t.data_ptr())


def default_schema_from_library() -> CSchema:
    s = CSchema()
    rc = _native.load().surge_replay_default_schema(ctypes.byref(s))
    if rc != 0:
        raise ReplayError(rc, "surge_replay_default_schema")
    return s


class ReplayEngine:
    """One handle = one GPU = one shard of the event log."""

    def __init__(self, algebra=DEFAULT_ALGEBRA, device: int = 0):
        """``algebra``: an ``EventAlgebra`` (ABI v1: the seven named fields) or a ``SlotAlgebra`` (ABI v2: typed slots)."""
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        self.algebra = algebra
        self.device = device
        sc = algebra.to_c()
        self.v2 = getattr(sc, "abi_version", 1) == 2  # a SlotAlgebra (ABI v2 slot schema)
        create = self._lib.surge_replay_create_v2 if self.v2 else self._lib.surge_replay_create
        rc = create(ctypes.byref(sc), device, ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.surge_replay_last_error(None)
            raise ReplayError(rc, msg.decode() if msg else "surge_replay_create failed")
        self.n_agg = 0
        self.state_dtype = algebra.state_dtype() if self.v2 else STATE_DTYPE  # numpy view of one 64-byte state
        self._keep = []  # tensors bound zero-copy must outlive the binding

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.surge_replay_destroy(self._h)
            self._h = ctypes.c_void_p()
            self._keep = []

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.surge_replay_last_error(self._h)
            raise ReplayError(rc, msg.decode() if msg else "")

    def use_stream(self, stream) -> None:
        """Launch on a ``torch.cuda.Stream`` (or ``None`` for the default stream)."""
        ptr = None if stream is None else ctypes.c_void_p(stream.cuda_stream)
        self._check(self._lib.surge_replay_set_stream(self._h, ptr))

    @property
    def stream_ptr(self) -> int:
        """The ``hipStream_t`` the engine's work goes to, as an integer (0 = the default stream)."""
        p = ctypes.c_void_p()
        self._check(self._lib.surge_replay_get_stream(self._h, ctypes.byref(p)))
        return int(p.value or 0)

    @contextlib.contextmanager
    def on_own_stream(self):
        """For the duration of the block the engine works on a non-blocking stream of its own (unless it was given one
        already): work that other streams enqueue — a device decoder's interning on the default stream — then overlaps the
        engine's group-by and fold instead of queueing behind them.  Synchronised on the way in and on the way out."""
        if self.stream_ptr:
            yield
            return
        import torch

        self.synchronize()
        st = torch.cuda.Stream(device=self.device)  # (PyTorch's pool streams are created non-blocking)
        self.use_stream(st)
        try:
            yield
        finally:
            try:
                self.synchronize()
            finally:
                self.use_stream(None)

    def synchronize(self) -> None:
        self._check(self._lib.surge_replay_synchronize(self._h))

    # -- load ---------------------------------------------------------------------------------
    def load_csr(self, seg_off, events, init_state=None, state_out=None) -> None:
        """Bind one shard's CSR log.  numpy arrays are copied H2D; CUDA tensors are bound in place."""
        if _is_torch(events) or _is_torch(seg_off):
            import torch

            if seg_off.dtype != torch.int64:
                raise ValueError("seg_off must be int64")
            n_agg = seg_off.numel() - 1
            n_events = (events.numel() * events.element_size()) // 16
            self._keep = [seg_off, events, init_state, state_out]
            self._check(
                self._lib.surge_replay_bind_device_csr(
                    self._h,
                    _dev_ptr(seg_off),
                    n_agg,
                    _dev_ptr(events),
                    n_events,
                    _dev_ptr(init_state, n_agg * 64),
                    _dev_ptr(state_out, n_agg * 64),
                )
            )
        else:
            seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
            events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
            n_agg = seg_off.shape[0] - 1
            if n_agg < 0:
                raise ValueError("seg_off needs at least one entry")
            if init_state is not None:
                init_state = np.ascontiguousarray(init_state, dtype=self.state_dtype)
                if init_state.shape[0] != n_agg:
                    raise ValueError("init_state must have one entry per aggregate")
            self._keep = []
            self._check(
                self._lib.surge_replay_load_csr(
                    self._h, _np_ptr(seg_off), n_agg, _np_ptr(events), events.shape[0], _np_ptr(init_state)
                )
            )
        self.n_agg = n_agg

    # -- fold ------------------------------------------------------------------------------------
    def fold(self, algo: int = ALGO_AUTO) -> None:
        self._check(self._lib.surge_replay_fold(self._h, algo))

    def prepare(self, algo: int = ALGO_AUTO) -> None:
        """Build the per-log index ``algo`` needs (length order, chunk table, tile-major copy) without folding."""
        self._check(self._lib.surge_replay_prepare(self._h, algo))

    def layout_info(self) -> CLayoutInfo:
        """The index built for the bound log and its one-off cost (``surge_replay_layout_info``)."""
        info = CLayoutInfo()
        self._check(self._lib.surge_replay_layout_info(self._h, ctypes.byref(info)))
        return info

    def index_order(self, algo: int) -> np.ndarray:
        """The row order of the bound log's index (``surge_replay_index_order``; diagnostics / tests)."""
        n = ctypes.c_int64()
        self._check(self._lib.surge_replay_index_order(self._h, algo, None, 0, ctypes.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.int64)
        self._check(self._lib.surge_replay_index_order(self._h, algo, out.ctypes.data_as(ctypes.c_void_p), out.shape[0], ctypes.byref(n)))
        return out[: int(n.value)]

    def kernel_info(self) -> dict:
        """Which build of the fold kernels this handle runs (``surge_replay_kernel_info``): v2 handles run kernels
        compiled for their schema at create time when libhiprtc is present."""
        info = CKernelInfo()
        self._check(self._lib.surge_replay_kernel_info(self._h, ctypes.byref(info)))
        return {"specialised": bool(info.specialised), "compile_ms": float(info.compile_ms), "detail": info.detail.decode(errors="replace")}

    def append_fold(self, group_agg, group_off, events) -> None:
        """Micro-batch re-fold onto the resident state (K3); see ``surge_replay_append_fold``."""
        if _is_torch(events):
            n_groups = group_agg.numel()
            n_events = (events.numel() * events.element_size()) // 16
            self._check(
                self._lib.surge_replay_append_fold_device(
                    self._h, _dev_ptr(group_agg), _dev_ptr(group_off), n_groups, _dev_ptr(events), n_events
                )
            )
            self._keep_batch = [group_agg, group_off, events]
        else:
            group_agg = np.ascontiguousarray(group_agg, dtype=np.int64)
            group_off = np.ascontiguousarray(group_off, dtype=np.int64)
            events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
            if group_off.shape[0] != group_agg.shape[0] + 1:
                raise ValueError("group_off needs n_groups + 1 entries")
            self._check(
                self._lib.surge_replay_append_fold(
                    self._h, _np_ptr(group_agg), _np_ptr(group_off), group_agg.shape[0], _np_ptr(events), events.shape[0]
                )
            )

    def append_events(self, agg_idx, events) -> None:
        """Micro-batch in topic order, event i tagged with ``agg_idx[i]``; grouping happens in the library, on the
        device.  CUDA tensors are taken in place, numpy arrays are staged through pinned memory."""
        if _is_torch(events):
            n_events = (events.numel() * events.element_size()) // 16
            if agg_idx.numel() != n_events:
                raise ValueError("one aggregate index per event")
            self._check(self._lib.surge_replay_append_events_device(self._h, _dev_ptr(agg_idx), _dev_ptr(events), n_events))
            self._keep_batch = [agg_idx, events]
            return
        agg_idx = np.ascontiguousarray(agg_idx, dtype=np.int64)
        events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
        if agg_idx.shape[0] != events.shape[0]:
            raise ValueError("one aggregate index per event")
        self._check(self._lib.surge_replay_append_events(self._h, _np_ptr(agg_idx), _np_ptr(events), events.shape[0]))

    # -- the device packer: decoded events in topic order -> a bound CSR log ---------------------
    def stage_reserve(self, n_events: int) -> None:
        self._check(self._lib.surge_replay_stage_reserve(self._h, int(n_events)))

    def stage_events(self, agg_idx, events) -> None:
        """Append CUDA tensors ``agg_idx`` (int64) / ``events`` (16 B each), topic order, to the staging log
        (``surge_replay_stage_events_device``); ``pack_staged`` turns everything staged into the bound log."""
        n_events = (events.numel() * events.element_size()) // 16
        if agg_idx.numel() != n_events:
            raise ValueError("one aggregate index per event")
        self._check(self._lib.surge_replay_stage_events_device(self._h, _dev_ptr(agg_idx), _dev_ptr(events), n_events))
        self._keep_batch = [agg_idx, events]

    @property
    def staged(self) -> int:
        n = ctypes.c_int64()
        self._check(self._lib.surge_replay_staged(self._h, ctypes.byref(n)))
        return int(n.value)

    def pack_staged(self, n_agg: int) -> None:
        """Everything staged -> a CSR log over ``n_agg`` aggregates, bound to the handle (``surge_replay_pack_staged``)."""
        self._check(self._lib.surge_replay_pack_staged(self._h, int(n_agg)))
        self.n_agg = int(n_agg)
        self._keep = []
        self._keep_batch = None

    def bound_log(self):
        """``(seg_off, events[n, 2] int64)`` CUDA tensor views of the bound log (valid until the next load / bind / pack)."""
        import torch

        po, pe = ctypes.c_void_p(), ctypes.c_void_p()
        na, ne = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.surge_replay_bound_log(self._h, ctypes.byref(po), ctypes.byref(pe), ctypes.byref(na), ctypes.byref(ne)))
        dev = torch.device("cuda", self.device)

        def view(ptr, shape):
            iface = {"shape": shape, "typestr": "<i8", "data": (ptr.value or 0, False), "version": 2}
            return torch.as_tensor(type("_Span", (), {"__cuda_array_interface__": iface})(), device=dev)

        so = view(po, (na.value + 1,))
        ev = view(pe, (ne.value, 2)) if ne.value else torch.zeros((0, 2), dtype=torch.int64, device=dev)
        return so, ev

    def grow(self, new_n_agg: int) -> None:
        """Extend the resident state to ``new_n_agg`` aggregates (the new ones ``None``); see ``surge_replay_grow``."""
        self._check(self._lib.surge_replay_grow(self._h, int(new_n_agg)))
        self.n_agg = max(self.n_agg, int(new_n_agg))

    # -- read --------------------------------------------------------------------------------------
    def snapshot(self) -> np.ndarray:
        out = np.zeros(self.n_agg, dtype=self.state_dtype)
        self._check(self._lib.surge_replay_snapshot(self._h, _np_ptr(out), None))
        return out

    def get(self, agg_idx: int) -> Optional[np.ndarray]:
        """Fixed-width state of one aggregate, or ``None`` when it is absent (KTable miss)."""
        out = np.zeros(1, dtype=self.state_dtype)
        present = ctypes.c_uint8(0)
        self._check(self._lib.surge_replay_get(self._h, int(agg_idx), _np_ptr(out), ctypes.byref(present)))
        return out[0] if present.value else None

    def get_raw(self, agg_idx: int) -> np.ndarray:
        out = np.zeros(1, dtype=self.state_dtype)
        self._check(self._lib.surge_replay_get(self._h, int(agg_idx), _np_ptr(out), None))
        return out[0]

    def gather(self, agg_idx) -> np.ndarray:
        """Fixed-width states of the listed aggregates (bulk point read)."""
        idx = np.ascontiguousarray(agg_idx, dtype=np.int64)
        out = np.zeros(idx.shape[0], dtype=self.state_dtype)
        self._check(self._lib.surge_replay_gather(self._h, _np_ptr(idx), idx.shape[0], _np_ptr(out)))
        return out

    def device_state(self):
        """The resident ``n_agg x 64`` byte state array as a ``torch.uint8`` view (no copy)."""
        import torch

        ptr = ctypes.c_void_p()
        n = ctypes.c_int64()
        self._check(self._lib.surge_replay_device_state(self._h, ctypes.byref(ptr), ctypes.byref(n)))
        for t in self._keep:
            if t is not None and _is_torch(t) and t.data_ptr() == ptr.value:
                return t.view(torch.uint8).reshape(-1)[: n.value * 64].view(n.value, 64)
        # handle-owned buffer: wrap through the CUDA array interface
        iface = {"shape": (n.value, 64), "typestr": "|u1", "data": (ptr.value or 0, False), "version": 3}
        holder = type("_DevView", (), {"__cuda_array_interface__": iface})()
        return torch.as_tensor(holder, device=f"cuda:{self.device}")

    def pack_states(self, states64, packed40, stream=None) -> None:
        """``n x 64`` -> ``n x 40`` bytes (snapshot wire form), on ``stream`` (default: the engine's)."""
        n = states64.shape[0]
        sp = None if stream is None else ctypes.c_void_p(stream.cuda_stream)
        self._check(self._lib.surge_replay_pack_states(self._h, _dev_ptr(states64, n * 64), n, _dev_ptr(packed40, n * 40), sp))

    def unpack_states(self, packed40, states64, stream=None) -> None:
        n = states64.shape[0]
        sp = None if stream is None else ctypes.c_void_p(stream.cuda_stream)
        self._check(self._lib.surge_replay_unpack_states(self._h, _dev_ptr(packed40, n * 40), n, _dev_ptr(states64, n * 64), sp))

    def set_state_out(self, tensor) -> None:
        """Redirect the next folds' output to ``tensor`` (``n_agg x 64`` bytes on the device)."""
        self._check(self._lib.surge_replay_set_state_out(self._h, _dev_ptr(tensor, self.n_agg * 64)))
        self._keep.append(tensor)
        if len(self._keep) > 8:
            del self._keep[4]

    # -- multi-GPU exchange (RCCL behind the C ABI) ----------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes rank 0 creates and hands to every rank (``surge_replay_comm_unique_id``)."""
        buf = (ctypes.c_uint8 * 128)()
        lib = _native.load()
        rc = lib.surge_replay_comm_unique_id(buf)
        if rc != 0:
            msg = lib.surge_replay_last_error(None)
            raise ReplayError(rc, msg.decode() if msg else "surge_replay_comm_unique_id failed")
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes) -> None:
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.surge_replay_comm_init(self._h, rank, world, buf))
        self.comm_rank, self.comm_world = rank, world

    def comm_destroy(self) -> None:
        self._check(self._lib.surge_replay_comm_destroy(self._h))

    def comm_info(self) -> dict:
        r, w, v = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        lib = ctypes.c_char_p()
        self._check(self._lib.surge_replay_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(v), ctypes.byref(lib)))
        return {"rank": r.value, "world": w.value, "rccl_version": v.value, "library": (lib.value or b"").decode()}

    def comm_counts(self, n_local: int):
        """``(counts per rank, max_count)`` for shards of ``n_local`` states on this rank (collective)."""
        counts = np.zeros(self.comm_world, dtype=np.int64)
        mx = ctypes.c_int64()
        self._check(self._lib.surge_replay_comm_counts(self._h, int(n_local), _np_ptr(counts), ctypes.byref(mx)))
        return counts, mx.value

    def allgather_snapshot(self, states, n_local: int, out, rows_per_rank: int, slot: int = 0, mode: int = 0) -> None:
        """``out[r, i] = state i of rank r`` for every rank (``surge_replay_allgather_snapshot``); asynchronous."""
        self._check(self._lib.surge_replay_allgather_snapshot(
            self._h, _dev_ptr(states, n_local * 64) if states is not None else None, int(n_local),
            _dev_ptr(out, self.comm_world * rows_per_rank * 64), int(rows_per_rank), slot, mode))

    @staticmethod
    def allgather_group(engines, n_local=None, outs=None, rows_per_rank: int = 0, slot: int = 0) -> None:
        """One host process, several handles (one per GPU): ``surge_replay_allgather`` — every engine ends up with every
        engine's resident states (peer copies, no RCCL).  ``outs`` = one device tensor per engine, or None: each engine
        keeps the gathered snapshot (``gathered_read``)."""
        n = len(engines)
        hs = (ctypes.c_void_p * n)(*[e._h for e in engines])
        cnt = None if n_local is None else np.ascontiguousarray(n_local, dtype=np.int64)
        if cnt is not None and cnt.shape != (n,):
            raise ValueError("n_local holds one count per engine")
        ptrs = None
        if outs is not None:
            ptrs = (ctypes.c_void_p * n)(*[_dev_ptr(o, n * rows_per_rank * 64) for o in outs])
        rc = engines[0]._lib.surge_replay_allgather(hs, n, _np_ptr(cnt), ptrs, int(rows_per_rank), slot)
        engines[0]._check(rc)
        for r, e in enumerate(engines):
            e.comm_rank, e.comm_world = r, n

    def gathered_read(self, slot: int, rank: int, first_row: int, n_rows: int) -> np.ndarray:
        """Rows of rank ``rank``'s block of the handle-owned gathered snapshot (waits for the slot's exchange)."""
        out = np.zeros(n_rows, dtype=self.state_dtype)
        self._check(self._lib.surge_replay_gathered_read(self._h, slot, rank, int(first_row), int(n_rows), _np_ptr(out)))
        return out

    def comm_wait(self, slot: int, host_sync: bool = False) -> None:
        self._check(self._lib.surge_replay_comm_wait(self._h, slot, 1 if host_sync else 0))

    # -- measurement ----------------------------------------------------------------------------------
    def stats_reset(self) -> None:
        self._check(self._lib.surge_replay_stats_reset(self._h))

    def fold_times_ms(self) -> np.ndarray:
        """HIP-event time of the dominant kernel of every fold since ``stats_reset`` (at most 256 are kept)."""
        out = np.zeros(256, dtype=np.float64)
        n = ctypes.c_int64()
        self._check(self._lib.surge_replay_fold_times(self._h, _np_ptr(out), 256, ctypes.byref(n)))
        return out[: n.value]

    def stats(self) -> CStats:
        st = CStats()
        self._check(self._lib.surge_replay_stats(self._h, ctypes.byref(st)))
        return st

    def stream_probe_ms(self, tensor) -> float:
        ms = ctypes.c_double(0.0)
        nbytes = (tensor.numel() * tensor.element_size()) // 16 * 16
        self._check(self._lib.surge_replay_stream_probe(self._h, _dev_ptr(tensor), nbytes, ctypes.byref(ms)))
        return ms.value

    def partition_hash_device(self, d_utf16, d_str_off, n_partitions: int, d_out, up_to_colon: bool = False) -> None:
        """K4: ``partitionForKey`` of whole strings (``KafkaPartitioner.scala:8``); ``up_to_colon`` first applies
        ``PartitionStringUpToColon.partitionBy`` (``:38-42``)."""
        n = d_str_off.numel() - 1
        fn = self._lib.surge_replay_partition_hash_up_to_colon_device if up_to_colon else self._lib.surge_replay_partition_hash_device
        self._check(fn(self._h, _dev_ptr(d_utf16), _dev_ptr(d_str_off), n, n_partitions, _dev_ptr(d_out)))
