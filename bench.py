#!/usr/bin/env python3
"""bench.py — events/sec replayed by the GPU aggregate fold (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload at every N (the log BASELINE.json's target is quoted on): the 10,000,000-aggregate synthetic log with Zipf(1..4096)
event counts (~4.6e9 events, 74 GB, 16 B events, 64 B state), ids ``acct-%08d``, STRONG-scaled: sharded
by the reference's own shard map — partition = partitionForKey(id, 64) (KafkaPartitioner.scala:8), gpu = partition % N
(ownership like PartitionAssignments.scala:51-63) — every rank generating only its shard.  N = 1 is BASELINE config C3 (the
whole log on one GPU: it fits), N = 8 is config C4.  One "step" = one full replay of the rank's HBM-resident shard and,
for N > 1, the all-gather of the final snapshot through the C ABI (RCCL inside libsurge_replay.so: grouped per-peer
send/recv of the 40-byte wire form on the library's side stream), overlapped with the next step's fold.  Inputs are
resident in HBM before the timed region.  ``--workload c2`` runs BASELINE config C2 (1 M aggregates x 256 events) instead;
``--workload c2-weak`` is round 1's weak-scaled C2-per-GPU run.

The default algorithm is what a one-shot recovery runs: AUTO — the fold straight from the CSR log, no copy of it (the
10 M-aggregate Zipf log: SORTED; C2: ROWS); ``one_shot`` carries its index build and first (cold) fold.  At N = 1 the line
also carries ``tile_major`` — the same log through the tile-major copy (SURGE_ALGO_TILED: group-major 8 KiB subtiles
streamed linearly), with what that copy costs once — ``secondary`` (config C2, both transports), ``c5`` (config 5, bounded),
``v2`` (the ABI v2 slot path), ``c4_shard`` (what ONE GPU of config C4 folds: rank 0's 1.25 M aggregates of the 8-GPU split,
AUTO, 100 folds, every aggregate checked) and ``e2e`` (events-topic BYTES -> states over 3e7 records of the same population on
a topic shaped like the reference's publisher writes it — one lz4 transaction + COMMIT marker per flush per partition, written
by the independent test-side producer — with the two neighbouring topic layouts beside it; states checked against the oracle's
fold of the source events).  ``python bench.py --gpus N`` with N > 1 and no torchrun environment starts its own
ranks (torch.distributed.run on 127.0.0.1).  ``--workload e2e [--gpus N] [--txn-flush-events K]``: that path alone
(run_e2e's docstring); ``--workload c4-shard``: the shard alone.

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant fold kernel against the 8 TB/s HBM peak using the
algorithmic bytes 16*E + 8*(A+1) + 64*A and the kernel's HIP-event times measured inside the timed region on
the launch stream (mean for ``achieved``; min / median / max reported).  ``traffic`` comes from the rocprofv3 PMC passes
committed under profiles/ for exactly this kernel and shape (profiles/traffic_manifest.json; null when there is none or
the kernel sources changed since).  At N = 1 ``secondary`` carries config C2 with its own roofline, and ``cpu_baseline``
times the CPU restatement (oracle/, "port") on a bounded sample of the same log on this box's host cores; the GPU result
for that sample is checked bit-for-bit against it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL between processes needs this (already exported on the GPU boxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
N_AGGREGATES = 10_000_000
N_PARTITIONS = 64
ZIPF_SEED = 3
C2_AGGREGATES, C2_EVENTS, C2_SEED = 1_000_000, 256, 2


ALGO_NAMES = {"auto": 0, "fixed": 1, "flat": 2, "rows": 3, "sorted": 4, "chunked": 5, "tiled": 7, "short": 8}


def kernel_name(S, algo):
    return {S.ALGO_FIXED: "fold_kernel<FIXED,16>", S.ALGO_FLAT: "fold_kernel<FLAT,16>", S.ALGO_ROWS: "fold_rows_kernel<8>",
            S.ALGO_SORTED: "fold_sorted_kernel<16>" if os.environ.get("SURGE_REPLAY_SORTED_KERNEL") == "plain" else "fold_sorted_pf_kernel<16>", S.ALGO_CHUNKED: "fold_chunked_kernel<16> + chunk_stitch_kernel",
            S.ALGO_TILED: "fold_tiled_kernel<2> + chunk_stitch_kernel", S.ALGO_SHORT: "fold_short_kernel"}.get(algo, str(algo))


def algo_name(S, algo):
    return {S.ALGO_FIXED: "fixed", S.ALGO_FLAT: "flat", S.ALGO_ROWS: "rows", S.ALGO_SORTED: "sorted", S.ALGO_CHUNKED: "chunked",
            S.ALGO_TILED: "tiled", S.ALGO_SHORT: "short"}.get(algo, str(algo))


def parse_algo(text):
    if text is None:
        return None
    if text.lower() in ALGO_NAMES:
        return ALGO_NAMES[text.lower()]
    return int(text)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def thread_cpu_seconds():
    """{tid: (name, CPU seconds)} of every thread of this process — Python threads by their names, native ones (the framing pool
    names its own; the HIP runtime's are left with the process's) by /proc/<pid>/task/<tid>/comm — read from each thread's CPU-time
    clock (clock id (~tid << 3) | 6: CPUCLOCK_SCHED, per thread), which has the scheduler's resolution, not a tick's."""
    import threading
    names = {t.native_id: t.name for t in threading.enumerate()}
    out = {}
    for tid in os.listdir("/proc/self/task"):
        tid = int(tid)
        try:
            comm = open(f"/proc/self/task/{tid}/comm").read().strip()
            out[tid] = (names.get(tid, comm), time.clock_gettime(((~tid) << 3) | 6))
        except (OSError, ValueError):
            pass  # (the thread ended between the listing and the read)
    return out


def csrc_sha16(kernel=""):
    """Identity of the sources ``kernel`` is built from (profiles/traffic_manifest.json records it per entry): the two
    headers every fold shares plus the translation unit(s) of that kernel — a change to the chunked kernel does not make
    the rows kernel's counter profile stale."""
    files = ["fold_layout.h", "fold_device.h"]
    if "sorted_pf" in kernel or "chunked" in kernel:
        files += ["fold_chunk_device.h", "fold_lane_device.h", "fold_chunked.hip"]
    elif "rows" in kernel:
        files += ["fold_chunk_device.h", "fold_lane_device.h", "fold_kernels.hip"]
    elif "tiled" in kernel and "slots" not in kernel:
        files += ["fold_chunk_device.h", "fold_tiled.hip"]
    elif "slots" in kernel:
        files += ["fold_slots_device.h", "fold_slots.hip"]
    elif "FLAT" in kernel:
        files += ["fold_flat_device.h", "fold_kernels.hip"]
    else:  # rows, sorted (plain), fixed
        files += ["fold_kernels.hip"]
    h = hashlib.sha256()
    d = os.path.join(ROOT, "surge_amd", "csrc")
    for name in files:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, algorithmic_bytes):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ for THIS kernel and THIS shape
    (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, gfx950 x2 correction on FETCH_SIZE, KB -> bytes; written by
    scripts/prof_traffic.py).  bench.py cannot collect PMC counters itself: null when no profile matches, and null —
    saying so — when the kernel sources changed after the profile was taken."""
    path = os.path.join(ROOT, "profiles", "traffic_manifest.json")
    if not os.path.exists(path):
        return None, None
    try:
        entries = json.load(open(path))
    except Exception:  # pragma: no cover
        return None, None
    sha = csrc_sha16(kernel)
    stale = None
    for e in entries:
        if e.get("kernel") == kernel and int(e.get("algorithmic_bytes", -1)) == int(algorithmic_bytes):
            if e.get("csrc_sha16") == sha:
                return e["traffic_bytes"], e.get("source")
            stale = e.get("source")
    if stale:
        return None, f"stale: kernel sources changed since {stale}"
    return None, None


class _LazyGlobalOffsets:
    """``global_seg_off[agg]`` for the sharded Zipf log without materialising the global prefix sum: only the event
    *hash index* needs to be unique per (aggregate, position), so aggregate a's events are numbered from a * 4096
    (4096 = the longest possible segment)."""

    def __getitem__(self, agg):
        return agg * 4096


def time_folds(eng, torch, dev, algo, steps, warmup):
    """K timed folds of the bound log on the engine's stream (no exchange); returns (elapsed_s, stats, per-launch ms)."""
    for _ in range(warmup):
        eng.fold(algo)
    torch.cuda.synchronize(dev)
    eng.stats_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.fold(algo)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return dt, eng.stats(), eng.fold_times_ms()


def roofline_of(S, st, times_ms, probe_gbps=None):
    import numpy as np

    kernel_ms = float(np.mean(times_ms)) if len(times_ms) else 0.0
    achieved = st.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    kname = kernel_name(S, st.last_algo)
    traffic, traffic_src = pmc_traffic(kname, st.algorithmic_bytes)
    r = {
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "kernel": kname,
        "kernel_ms": kernel_ms,
        "kernel_ms_min_median_max": [float(np.min(times_ms)), float(np.median(times_ms)), float(np.max(times_ms))] if len(times_ms) else None,
        "algorithmic_bytes": st.algorithmic_bytes,
        "timed_launches": int(len(times_ms)),
    }
    if probe_gbps is not None:
        r["stream_read_probe_GBps"] = probe_gbps
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=["c4", "c3", "c2", "c2-weak", "c5", "v2", "e2e", "c4-shard"],
                    help="c4 / c3 (default): the 10 M-aggregate Zipf log, strong-scaled over the GPUs; c2: 1 M x 256 fixed fan-in "
                         "(strong-scaled); c2-weak: 1 M x 256 PER GPU (round 1's run); c5: streaming micro-batches onto a resident "
                         "state store with periodic state-topic snapshots (one GPU; a step = one micro-batch, default 600 steps); "
                         "v2: an ABI v2 slot schema (accumulating f64 ledger) over a 2 M-aggregate Zipf log, schema-specialised "
                         "kernels vs the generic interpreter, CSR vs tile-major transport (one GPU); e2e: events-topic BYTES "
                         "(Kafka record batches of play-json Counter events) -> host framing -> device decode -> device group-by + fold "
                         "-> states, one host thread (a step = one fetch of --batch-events records)")
    ap.add_argument("--batch-events", type=int, default=100_000, help="c5: events per micro-batch")
    ap.add_argument("--snapshot-every", type=int, default=30, help="c5: publish a state-topic delta every N batches (0 = never)")
    ap.add_argument("--device-batches", action="store_true", help="c5: batches already in HBM (no staging / H2D)")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "none"], help="e2e: compression of the record batches (the reference's producer: lz4)")
    ap.add_argument("--host-framing", action="store_true", help="c5: frame the state-topic record batches with the host writer instead of the device framer")
    ap.add_argument("--consumer-waits", choices=("both", "finish", "none"), default="finish", help="e2e, one consumer thread: where it waits for the device — behind interning and behind the fold (both), "
                    "behind interning only (finish: the fold is handed over by an event and runs while the next push is enqueued), or only in the middle of interning (none)")
    ap.add_argument("--two-thread-consumer", action="store_true", help="e2e: a worker thread enqueues the pushes, this one finishes them with an event-ordered hand-over to the fold and no host wait (default: one thread does both, with the host waits --consumer-waits names — the device is the bound, the extra thread measured no faster at 1 ms more CPU per fetch)")
    ap.add_argument("--serial-framing", action="store_true", help="e2e: frame each fetch, then push it, then fold it, one after the other (default: framing one fetch ahead on its own threads, four pushes in flight)")
    ap.add_argument("--events-cap", type=int, default=8, help="e2e: every aggregate publishes the first min(count, cap) of its events (8: 6.4e7 records over the 10 M aggregates)")
    ap.add_argument("--e2e-topic", default="counter", choices=["counter", "mixed"],
                    help="e2e: counter = the Counter fixture's events on the C3 / C4 population (keys <id>:<seq>, Int arguments); mixed = the surge-docs BankAccount model's events "
                         "(BankAccountSurgeModel.scala:26-32: UUID record keys without ':', no sequence numbers, Double balances as play-json text), every record with two "
                         "headers (a W3C traceparent and a content type: SurgeModel.scala:46-52 passes a message's headers and tracing context on), and every 128th account "
                         "publishing 64..256 events instead of <= --events-cap — 4 M accounts by default")
    ap.add_argument("--writer", default="independent", choices=["independent", "product"],
                    help="e2e: who writes the topic — the independent test-side producer (tests/native/wire_writer.c: shares no code with the library that "
                         "reads it; default) or the product's own RecordBatchWriter (plain 16 KiB batches only)")
    ap.add_argument("--txn-flush-events", type=int, default=512,
                    help="e2e, independent writer: records per partition per publisher flush — every flush ONE transaction closed by a COMMIT control batch, "
                         "its data batches closed by the flush or at 16 KiB (KafkaProducerActorImpl.scala:397-453, flush-interval 50 ms: reference.conf:20); "
                         "0 = no transactions, every batch filled to 16 KiB (rounds 3 / 4's layout)")
    ap.add_argument("--abort-every", type=int, default=50, help="e2e, transactional topic: every N-th flush of a partition first fails (its records + an ABORT marker) and is retried; 0 = none")
    ap.add_argument("--hold-markers", type=int, default=4, help="e2e, transactional topic: on partitions p %% N == 1 the last marker of a fetch response arrives with the next one; 0 = never")
    ap.add_argument("--no-capacity-hint", action="store_true", help="e2e: let the resident state and the key table grow as aggregates appear instead of sizing them up front")
    ap.add_argument("--framing-by-copy", action="store_true", help="e2e: frame the way rounds 4 / 5 did — every records section copied into the framer's slab and its CRC-32C run "
                    "on the host (default: fetch responses are received into the slab and framed in place, the CRC-32C is finished on the device: the host reads batch headers only)")
    ap.add_argument("--framing-threads", type=int, default=2, help="e2e: host threads receiving and framing a fetch's partitions side by side (2: in-place framing with the CRC on the device leaves the host the "
                    "batch headers; rounds 4 / 5 framed by copy with the CRC on the host and needed 12 of the boxes' 16-CPU quota — --framing-by-copy --framing-threads 12; "
                    "capped at this rank's share of the CPUs the process may use)")
    ap.add_argument("--host-only", action="store_true", help="e2e: the HOST side alone — every rank receives and frames its fetch responses (its share of the framing threads, "
                    "page-locked slabs) and hands nothing to the device: the rate the host sustains, its CPU time and its page-locked bytes per rank.  With --gpus 8 on a one-GPU box "
                    "(SURGE_BENCH_REHEARSAL=1): what bounds an 8-rank node on the host side, which no one-GPU run of the whole path can show")
    ap.add_argument("--bound-log", action="store_true", help="e2e: a recovery that folds ONCE — every fetch's decoded events are staged on the device (surge_replay_stage_decoded), the topic's "
                    "end packs them into one CSR log (surge_replay_pack_staged) and ONE fold of what AUTO picks for that log produces the states; the packed log is then re-folded a few times "
                    "(refold_events_per_s).  Default: every fetch is folded onto the resident state as it arrives (K3)")
    ap.add_argument("--aggregates", type=int, default=None, help="global aggregate count (default 10 M for c4, 1 M for c2)")
    ap.add_argument("--events-per-aggregate", type=int, default=C2_EVENTS, help="c2 only")
    ap.add_argument("--algo", default=None,
                    help="auto | fixed | flat | rows | sorted | chunked | tiled (or the SURGE_ALGO_* number); default: auto — the "
                         "kernel the engine picks for a log it folds straight from CSR (the tile-major fold is reported beside it at N = 1)")
    ap.add_argument("--parity", default="full", choices=["full", "sample", "none"],
                    help="N = 1: check the GPU states against the CPU restatement on the whole log (default), on the cpu_baseline "
                         "sample only, or not at all")
    ap.add_argument("--gather", default="native", choices=["native", "torch", "none"],
                    help="N > 1 snapshot exchange: native = RCCL behind the C ABI (default), torch = torch.distributed, none = fold only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-C2 secondary line at N = 1")
    args = ap.parse_args()
    if args.workload == "c5":
        print(json.dumps(run_c5(args)))
        return
    if args.workload == "v2":
        print(json.dumps(run_v2(args)))
        return
    if args.workload == "c4-shard":  # one GPU's share of config C4 (rank 0 of 8) on this GPU: the default line's c4_shard leg alone
        import torch

        from surge_amd import schema as S
        from surge_amd import synth
        from surge_amd.replay import ReplayEngine

        world, rank, local_rank, dev, ctl, dist, rehearsal = init_ranks(args, torch)
        print(json.dumps(run_c4_shard(args, S, synth, ReplayEngine, torch, dev, local_rank, steps=args.steps if args.steps != 20 else 100,
                                      algo=parse_algo(args.algo), check=not args.no_cpu_baseline)))
        return
    if args.workload == "e2e":
        line = run_e2e(args)
        if line is not None:  # rank 0
            print(json.dumps(line))
        return

    import numpy as np
    import torch

    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.replay import ReplayEngine

    world, rank, local_rank, dev, ctl, dist, rehearsal = init_ranks(args, torch)

    zipf = args.workload in ("c4", "c3")
    weak = args.workload == "c2-weak"
    algo = parse_algo(args.algo)
    if algo is None:
        algo = S.ALGO_AUTO  # what a recovery runs: the fold straight from the CSR log, no copy of it (round 3's default was TILED)
    L = args.events_per_aggregate
    n_global = args.aggregates or (N_AGGREGATES if zipf else C2_AGGREGATES)
    if weak:
        n_global *= world
    eng = ReplayEngine(device=local_rank)
    compute = torch.cuda.Stream(device=dev)
    eng.use_stream(compute)

    # ---- this rank's HBM-resident shard ---------------------------------------------------------------
    t_gen = time.perf_counter()
    if world == 1:
        agg_ids = torch.arange(n_global, dtype=torch.int64, device=dev)
    else:
        from surge_amd.dist import local_aggregate_ids

        agg_ids = local_aggregate_ids(n_global, N_PARTITIONS, rank, world, dev, eng)
    n_local = int(agg_ids.numel())
    if zipf:
        # the rank's aggregates keep the event counts and contents they have in the global log
        lens = synth.zipf_lengths(agg_ids, ZIPF_SEED)
        seg_off, events = synth.csr_log_device(lens, ZIPF_SEED, agg_ids=agg_ids, global_seg_off=_LazyGlobalOffsets())
        n_events_local = int(seg_off[-1].item())
        del lens
    elif world == 1:
        seg_off, events = synth.fixed_log_device(n_global, L, C2_SEED, dev)
        n_events_local = n_local * L
    else:
        seg_off, events = synth.fixed_log_for_aggregates_device(agg_ids, L, C2_SEED)
        n_events_local = n_local * L
    torch.cuda.synchronize(dev)
    gen_s = time.perf_counter() - t_gen

    # ---- the exchange (N > 1) -------------------------------------------------------------------------
    gather, gather_kind = None, None
    if world > 1 and args.gather != "none":
        from surge_amd.dist import NativeSnapshotGather, SnapshotGather

        ok = torch.ones(1, dtype=torch.int32, device=ctl)
        if args.gather == "native":
            try:
                gather = NativeSnapshotGather(n_local, dev, eng)
                gather_kind = "C ABI (surge_replay_allgather_snapshot): RCCL inside libsurge_replay.so"
            except Exception as exc:  # pragma: no cover - depends on the box
                print(f"[bench] rank {rank}: native exchange unavailable: {exc}", file=sys.stderr)
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if args.gather == "torch" or int(ok.item()) == 0:
            if gather is not None:
                eng.comm_destroy()
            gather = SnapshotGather(n_local, dev, engine=eng)
            gather_kind = "torch.distributed (nccl = RCCL)" + (" — FALLBACK: the C-ABI exchange failed to initialise" if args.gather == "native" else "")
    if gather is not None:
        bufs = gather.make_local_buffers()
    else:
        bufs = [torch.zeros((n_local, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
    eng.load_csr(seg_off, events, None, bufs[0])
    fold_done = [torch.cuda.Event(), torch.cuda.Event()]

    # ---- what happens ONCE per bound log: its index (length order / chunk table / tile-major copy) and the first,
    # cold fold — a recovery is one fold, so these are reported beside the warm-replay rate, never inside it
    torch.cuda.synchronize(dev)
    t_p = time.perf_counter()
    eng.prepare(algo)
    eng.synchronize()
    prepare_wall_ms = (time.perf_counter() - t_p) * 1e3
    layout = eng.layout_info()
    t_p = time.perf_counter()
    eng.fold(algo)
    eng.synchronize()
    first_fold_wall_ms = (time.perf_counter() - t_p) * 1e3
    first_fold_kernel_ms = eng.stats().last_fold_kernel_ms

    def step(i):
        slot = i & 1
        if gather is not None:
            gather.wait(slot, compute)  # the exchange that last read bufs[slot] must be finished
        eng.set_state_out(bufs[slot])
        eng.fold(algo)
        if gather is not None:
            fold_done[slot].record(compute)
            gather.launch(slot, bufs[slot], fold_done[slot])

    def sync_all():
        torch.cuda.synchronize(dev)  # every stream of the device, the library's side stream included
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    sync_all()
    eng.stats_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    t1 = time.perf_counter()
    last = (args.warmup + args.steps - 1) & 1

    st = eng.stats()
    times_ms = eng.fold_times_ms()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=ctl)
    totals = torch.tensor([n_events_local, n_local], dtype=torch.int64, device=ctl)
    per_rank = torch.tensor([float(n_events_local), float(np.mean(times_ms)) if len(times_ms) else 0.0], dtype=torch.float64, device=ctl)
    exchange = None
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        allr = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
        per_rank_events = [int(t[0].item()) for t in allr]
        per_rank_fold_ms = [float(t[1].item()) for t in allr]
        if gather is not None:
            # (a) the gathered snapshot holds every rank's shard: checksum of each rank's block against that rank's own
            torch.cuda.synchronize(dev)
            res = gather.result(last)
            own = torch.stack([bufs[last][: n_local].view(torch.int64).sum()]).to(ctl)
            sums = [torch.zeros_like(own) for _ in range(world)]
            dist.all_gather(sums, own)
            for r in range(world):
                got = res[r, : gather.counts[r]].reshape(-1).view(torch.int64).sum()
                assert int(got.item()) == int(sums[r].item()), f"rank {rank}: gathered block of rank {r} differs from its shard"
            assert torch.equal(res[rank, :n_local], bufs[last][:n_local]), "all-gathered snapshot does not contain the local shard"
            # (b) the exchange alone, not overlapped (barrier, launch, wait)
            ex = []
            for _ in range(5):
                sync_all()
                ta = time.perf_counter()
                gather.launch(last, bufs[last], None)
                torch.cuda.synchronize(dev)
                ex.append((time.perf_counter() - ta) * 1e3)
            exm = torch.tensor([float(np.median(ex))], dtype=torch.float64, device=ctl)
            dist.all_reduce(exm, op=dist.ReduceOp.MAX)
            exchange = float(exm.item())
    else:
        per_rank_events, per_rank_fold_ms = [n_events_local], [float(np.mean(times_ms)) if len(times_ms) else 0.0]
    elapsed_s = float(elapsed.item())
    total_events, total_aggs = int(totals[0].item()), int(totals[1].item())

    result = None
    if rank == 0:
        probe_gbps = None
        try:
            ms = min(eng.stream_probe_ms(events[: min(events.shape[0], 1 << 29)]) for _ in range(5))  # an 8 GB window, best of five
            probe_gbps = min(events.shape[0], 1 << 29) * 16 / (ms * 1e-3) / 1e9
        except Exception as e:  # pragma: no cover
            print(f"stream probe failed: {e}", file=sys.stderr)
        ms_per_step = elapsed_s / args.steps * 1e3
        roof = roofline_of(S, st, times_ms, probe_gbps)
        # what ONE recovery pays: the bound log's index + its first (cold) fold — a recovery folds once, the warm-replay rate
        # above is what a re-fold of the resident log runs at (device time between HIP events; *_wall: host clock, allocations included)
        if layout.index_build_ms + first_fold_kernel_ms > 0:
            roof["frac_one_shot"] = st.algorithmic_bytes / ((layout.index_build_ms + first_fold_kernel_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS
            roof["frac_one_shot_wall"] = st.algorithmic_bytes / ((prepare_wall_ms + first_fold_wall_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS
            roof["one_shot_ms"] = {"index_build": layout.index_build_ms, "first_fold_kernel": first_fold_kernel_ms, "prepare_wall": prepare_wall_ms,
                                   "first_fold_wall": first_fold_wall_ms}
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = run_cpu_baseline(args, seg_off, events, bufs[last])
        one_shot = {
            "index_algo": algo_name(S, layout.algo) if layout.algo else None,
            "index_build_ms": layout.index_build_ms,
            "relayout_ms": layout.relayout_ms,
            "prepare_wall_ms": prepare_wall_ms,
            "first_fold_kernel_ms": first_fold_kernel_ms,
            "first_fold_wall_ms": first_fold_wall_ms,
            "tile_major_copy_bytes": layout.tiled_bytes,
            "padding_events": layout.padding_events,
            "virtual_rows": layout.virtual_rows,
            "cut_aggregates": layout.cut_aggregates,
            "chunk_events": layout.chunk_events,
            "note": "device time between HIP events (wall for *_wall_ms), rank 0; paid once per bound log, outside the timed region",
        }
        # the other transport, measured in the same run on the same log (N = 1): the tile-major fold when the primary folds
        # straight from the CSR log (the default), the CSR fold when --algo tiled was asked for
        csr_direct, tile_major = None, None
        if world == 1 and st.last_algo == S.ALGO_TILED:
            eng.set_state_out(bufs[1 - last])
            t_p = time.perf_counter()
            eng.fold(S.ALGO_AUTO)
            eng.synchronize()
            csr_first_wall = (time.perf_counter() - t_p) * 1e3
            csr_layout = eng.layout_info()
            dt2, st2, tm2 = time_folds(eng, torch, dev, S.ALGO_AUTO, max(5, args.steps // 2), 1)
            csr_direct = {"algo": algo_name(S, st2.last_algo), "kernel": kernel_name(S, st2.last_algo),
                          "index_build_ms": csr_layout.index_build_ms, "first_fold_wall_ms_incl_index": csr_first_wall,
                          "kernel_ms_min_median_max": [float(np.min(tm2)), float(np.median(tm2)), float(np.max(tm2))],
                          "frac": st2.algorithmic_bytes / (float(np.mean(tm2)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                          "events_per_sec": total_events / (float(np.mean(tm2)) * 1e-3),
                          "states_equal_primary": bool(torch.equal(bufs[0], bufs[1]))}
            eng.set_state_out(bufs[last])
        elif world == 1 and zipf and not args.no_secondary:
            try:
                eng.set_state_out(bufs[1 - last])
                torch.cuda.synchronize(dev)
                t_p = time.perf_counter()
                eng.prepare(S.ALGO_TILED)  # a second copy of the log in HBM + one re-layout pass: paid once per bound log
                eng.synchronize()
                tm_prepare_wall = (time.perf_counter() - t_p) * 1e3
                tl = eng.layout_info()
                dt2, st2, tm2 = time_folds(eng, torch, dev, S.ALGO_TILED, max(5, args.steps // 2), 2)
                tile_major = {"algo": algo_name(S, st2.last_algo), "kernel": kernel_name(S, st2.last_algo),
                              "kernel_ms_min_median_max": [float(np.min(tm2)), float(np.median(tm2)), float(np.max(tm2))],
                              "frac": st2.algorithmic_bytes / (float(np.mean(tm2)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "events_per_sec": total_events / (float(np.mean(tm2)) * 1e-3),
                              "traffic": pmc_traffic(kernel_name(S, st2.last_algo), st2.algorithmic_bytes),
                              "one_shot": {"index_build_ms": tl.index_build_ms, "relayout_ms": tl.relayout_ms, "prepare_wall_ms": tm_prepare_wall,
                                           "tile_major_copy_bytes": tl.tiled_bytes, "padding_events": tl.padding_events,
                                           "note": "prepare_wall_ms includes a device allocation of the log's size: on a box whose free VRAM was handed back moments "
                                                   "ago that allocation alone can take seconds (the driver scrubs freed VRAM before it hands it out again: "
                                                   "profiles/r04_vram_alloc_probe.txt) — which is why the headline no longer depends on this copy"},
                              "states_equal_primary": bool(torch.equal(bufs[0], bufs[1]))}
                eng.set_state_out(bufs[last])
            except Exception as exc:  # pragma: no cover - e.g. not enough HBM for the copy
                tile_major = {"skipped": str(exc)}
        if zipf:
            wl = (f"C{'3' if world == 1 else '4'}: {n_global} aggregates, Zipf(1..4096) events each, CSR, 16 B events, 64 B state, "
                  f"ids acct-%08d sharded over {world} GPU(s) by partitionForKey(id, {N_PARTITIONS}) % {world}; log resident in HBM")
        else:
            wl = (f"C2{' per GPU (weak)' if weak else ''}: {n_global} aggregates x {L} events, 16 B events, 64 B state, sharded over {world} "
                  f"GPU(s); log resident in HBM")
        fold_ms = max(per_rank_fold_ms)
        result = {
            "metric": "events/sec replayed",
            "value": total_events * args.steps / elapsed_s,
            "unit": "events/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "int32/int64 adds + bit-copied f64",
            "data": "synthetic (counter-hash log; surge_amd/synth.py), generated on the device in %.0f s" % gen_s,
            **({"rehearsal": "every rank on cuda:0, gloo control plane: a functional run of the N > 1 path, not a measurement"} if rehearsal else {}),
            "aggregates_per_sec": total_aggs * args.steps / elapsed_s,
            "config": {
                "workload": wl,
                "aggregates": n_global,
                "events": total_events,
                "events_per_aggregate": "zipf(1..4096), mean ~460" if zipf else L,
                "algo": algo_name(S, st.last_algo),
                "wave_tasks": st.n_tasks,
                "per_rank_events": per_rank_events,
                "per_rank_fold_kernel_ms": per_rank_fold_ms,
                "exchange": None if world == 1 else (gather_kind or "none (fold only)"),
                "exchange_transport": None if gather is None else
                f"{'ncclAllGather, max-padded' if gather.mode == 'allgather' else 'grouped per-peer ncclSend/ncclRecv'}, "
                f"{'40-byte wire form' if gather.packed else '64-byte states'}, side stream, overlapped with the next fold",
                "exchange_alone_ms": exchange,
                "exchange_hidden_fraction": None if not exchange else max(0.0, min(1.0, 1.0 - max(0.0, ms_per_step - fold_ms) / exchange)),
            },
            "roofline": roof,
            "one_shot": one_shot,
            "csr_direct": csr_direct,
            "tile_major": tile_major,
            "cpu_baseline": cpu_baseline,
        }
    eng.close()
    del events, seg_off
    if rank == 0 and world == 1 and zipf and not args.no_secondary:
        torch.cuda.empty_cache()
        result["secondary"] = run_secondary_c2(args, S, synth, ReplayEngine, torch, dev, local_rank)
        # BASELINE config 5 and the ABI v2 path, bounded to a few seconds each, so that the driver's run times them too
        import argparse as _ap

        try:
            c5 = run_c5(_ap.Namespace(**{**vars(args), "workload": "c5", "steps": 150, "warmup": 3, "batch_events": 100_000, "snapshot_every": 30,
                                         "device_batches": False, "host_framing": False, "aggregates": None, "parity": "full"}))
            result["c5"] = {k: c5[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "cpu_baseline")}
        except Exception as exc:  # pragma: no cover
            result["c5"] = {"skipped": repr(exc)}
        try:
            v2 = run_v2(_ap.Namespace(**{**vars(args), "workload": "v2", "steps": 10, "warmup": 2, "aggregates": None}))
            result["v2"] = {k: v2[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "roofline", "one_shot", "cpu_baseline")}
        except Exception as exc:  # pragma: no cover
            result["v2"] = {"skipped": repr(exc)}
        # the kernel BASELINE config C4 runs on every GPU: rank 0's shard of the 8-GPU split (its partitions' aggregates), AUTO, 100 folds
        try:
            torch.cuda.empty_cache()
            result["c4_shard"] = run_c4_shard(args, S, synth, ReplayEngine, torch, dev, local_rank)
        except Exception as exc:  # pragma: no cover
            result["c4_shard"] = {"skipped": repr(exc)}
        # topic BYTES -> states (what a recovery really runs: SurgeStateStoreConsumer.scala:57-76), bounded to 3e7 records of the C3 population, on a
        # topic shaped like the reference's publisher writes it (one transaction + COMMIT marker per flush per partition) by the independent writer;
        # beside it the same path on the two neighbouring layouts (small flushes / no transactions and full 16 KiB batches), 1.2e7 records each
        try:
            torch.cuda.empty_cache()
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "data", "config", "roofline", "cpu_baseline")
            base = {**vars(args), "workload": "e2e", "warmup": 2, "batch_events": 1_000_000, "aggregates": None, "events_cap": 8, "writer": "independent",
                    "codec": "lz4", "serial_framing": False, "two_thread_consumer": False, "no_capacity_hint": False, "framing_threads": 2, "framing_by_copy": False, "consumer_waits": "finish",
                    "abort_every": 50, "hold_markers": 4, "bound_log": False, "e2e_topic": "counter"}
            # (the primary leg runs the first <= 14 events of every aggregate: 1.05e8 records, 103 timed fetches — a region long enough that one stalled fetch does not
            # decide the figure; the comparison legs below keep the 6.5e7-record topic)
            e2e = run_e2e(_ap.Namespace(**{**base, "steps": 200, "txn_flush_events": 512, "events_cap": 14}))
            result["e2e"] = {k: e2e[k] for k in keep}
            layouts = {"flush_512": {"value": e2e["value"], "control_batches": e2e["config"]["control_batches"], "parity": e2e["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}}
            for name, kf in (("flush_64", 64), ("full_16KiB_no_transactions", 0)):
                torch.cuda.empty_cache()
                o = run_e2e(_ap.Namespace(**{**base, "steps": 10, "txn_flush_events": kf}))
                layouts[name] = {"value": o["value"], "control_batches": o["config"]["control_batches"], "data_batches": o["config"]["topic"]["data_batches"],
                                 "events_timed": o["config"]["events_timed"], "parity": o["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}
            result["e2e"]["layouts_events_per_s"] = layouts
            # ... and the same topic the way a recovery that folds ONCE runs it: fetches staged on the device, one pack, one fold of the kernel AUTO picks
            # for the packed log (1.2e7 records: the lane-per-row kernels need a log of that size), re-folds of the packed log beside it
            torch.cuda.empty_cache()
            # — on a topic whose aggregates are LONG (250 000 aggregates of the same Zipf(1..4096) counts, every event of each: 1.1e8 records, 114 fetches), so that
            # what AUTO picks for the packed log is a lane-per-row kernel (CHUNKED), the family of the headline number
            o = run_e2e(_ap.Namespace(**{**base, "steps": 400, "txn_flush_events": 512, "bound_log": True, "aggregates": 250_000, "events_cap": 4096}))
            result["e2e"]["bound_log"] = {"topic": "250 000 aggregates, Zipf(1..4096) events each, all published", "fetches_timed": o["steps"],
                                          "value_bytes_to_staged_events_per_s": o["value"], **(o["config"]["bound_log"] or {}),
                                          "parity": o["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}
            # the same default topic framed the way rounds 4 / 5 framed it (sections copied into the slab, CRC-32C on 12 host threads): what in-place framing + the device CRC replace
            torch.cuda.empty_cache()
            o = run_e2e(_ap.Namespace(**{**base, "steps": 10, "txn_flush_events": 512, "framing_by_copy": True, "framing_threads": 12}))
            result["e2e"]["framing_by_copy_12_threads"] = {"value": o["value"], **{k: o["config"][k] for k in ("host_cpu_ms_per_1e6_records", "framing_cpu_ms_per_1e6_records", "framing_threads")},
                                                           "parity": o["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}
            # ... and in place with eight threads (the primary figure stays at two: what a rank of an 8-GPU node on a 16-CPU quota can afford)
            torch.cuda.empty_cache()
            o = run_e2e(_ap.Namespace(**{**base, "steps": 10, "txn_flush_events": 512, "framing_threads": 8}))
            result["e2e"]["in_place_8_threads"] = {"value": o["value"], **{k: o["config"][k] for k in ("host_cpu_ms_per_1e6_records", "framing_cpu_ms_per_1e6_records", "framing_threads")},
                                                   "parity": o["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}
            # the wider topic (--e2e-topic mixed): BankAccount events — UUID keys without ':', Double balances as text — with two headers per record and a slice of
            # accounts publishing 64..256 events; 4 M accounts, the whole topic
            try:
                torch.cuda.empty_cache()
                o = run_e2e(_ap.Namespace(**{**base, "steps": 100, "txn_flush_events": 512, "e2e_topic": "mixed"}))
                result["e2e"]["mixed_topic"] = {"value": o["value"], "workload": o["config"]["workload"], "fetches_timed": o["steps"], "events_timed": o["config"]["events_timed"],
                                                **{k: o["config"][k] for k in ("record_header_bytes", "max_events_of_one_aggregate", "wire_bytes_per_record", "host_cpu_ms_per_1e6_records",
                                                                               "host_cpu_ms_per_1e6_records_without_the_receive_copy", "keys_interned")},
                                                "doubles_parsed_on_host": o["config"]["decoder"]["doubles_parsed_on_host"],
                                                "parity": o["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"]}
            except Exception as exc:  # pragma: no cover
                result["e2e"]["mixed_topic"] = {"skipped": repr(exc)}
        except Exception as exc:  # pragma: no cover
            result["e2e"] = {"skipped": repr(exc)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def run_c4_shard(args, S, synth, ReplayEngine, torch, dev, local_rank, world=8, rank=0, steps=100, algo=None, check=True):
    """What ONE GPU of BASELINE config C4 folds — rank ``rank``'s shard of the 10 M-aggregate Zipf log split over ``world``
    GPUs by partitionForKey(id, 64) % world (KafkaPartitioner.scala:8, PartitionAssignments.scala:51-63): ~1.25 M aggregates,
    ~5.8e8 events — on this GPU, through AUTO, exactly as ``--gpus 8`` generates and folds it (no exchange: the driver has no
    8-GPU node; this times the kernel that configuration runs).  Every aggregate is checked against the CPU restatement."""
    import numpy as np

    from oracle import oracle
    from surge_amd.dist import local_aggregate_ids

    with ReplayEngine(device=local_rank) as e2:
        agg_ids = local_aggregate_ids(N_AGGREGATES, N_PARTITIONS, rank, world, dev, e2)
        lens = synth.zipf_lengths(agg_ids, ZIPF_SEED)
        so, ev = synth.csr_log_device(lens, ZIPF_SEED, agg_ids=agg_ids, global_seg_off=_LazyGlobalOffsets())
        n_local, n_ev = int(agg_ids.numel()), int(so[-1].item())
        out = torch.zeros((n_local, 64), dtype=torch.uint8, device=dev)
        e2.load_csr(so, ev, None, out)
        torch.cuda.synchronize(dev)
        t_p = time.perf_counter()
        algo = S.ALGO_AUTO if algo is None else algo
        e2.fold(algo)
        e2.synchronize()
        first_wall = (time.perf_counter() - t_p) * 1e3
        layout = e2.layout_info()
        dt, st, times_ms = time_folds(e2, torch, dev, algo, steps, 5)
        t_or = time.perf_counter()
        parity = None
        if check:
            exp = oracle.fold_csr(so.cpu().numpy(), synth.to_event_records(ev), threads=int(effective_cpus()[0]))
            got = out.cpu().numpy().view(S.STATE_DTYPE).reshape(-1)
            parity = bool(got.tobytes() == exp.tobytes())
        oracle_s = time.perf_counter() - t_or
        roof = roofline_of(S, st, times_ms)
        if first_wall > 0:  # one recovery of the shard: index + first fold, host clock
            roof["frac_one_shot_wall"] = st.algorithmic_bytes / (first_wall * 1e-3) / 1e9 / HBM_PEAK_GBPS
        return {
            "config": {"workload": f"C4, one GPU's share: rank {rank} of {world} — {n_local} aggregates ({n_ev} events) of the 10 M-aggregate Zipf(1..4096) log, "
                                   f"partitions p % {world} == {rank}; log resident in HBM; no exchange",
                       "algo": algo_name(S, st.last_algo), "wave_tasks": st.n_tasks, "aggregates": n_local, "events": n_ev},
            "metric": "events/sec replayed", "value": n_ev * steps / dt, "unit": "events/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "eight_gpu_projection_events_per_s": 8 * n_ev * steps / dt,
            "projection_note": "8 x this rate = what eight such GPUs fold per second when the snapshot exchange hides behind the next fold (bench.py --gpus 8 "
                               "measures that; no 8-GPU node was available to the builder) — a projection, not a measurement",
            "roofline": roof,
            "one_shot": {"index_build_ms": layout.index_build_ms, "first_fold_wall_ms_incl_index": first_wall},
            "cpu_baseline": {"gpu_matches_cpu_full_shard": parity, "aggregates_checked": n_local, "events_checked": n_ev, "seconds": oracle_s, "kind": "port"},
        }


def run_secondary_c2(args, S, synth, ReplayEngine, torch, dev, local_rank):
    """BASELINE config C2 (1 M aggregates x 256 events, uniform fan-in) on the same GPU, with its own roofline: the fold AUTO
    picks straight from the CSR log (ROWS) and, beside it, the tile-major fold with its one-off layout cost — each with the
    counter traffic of the committed PMC profile of exactly that kernel and shape."""
    import numpy as np

    so, ev = synth.fixed_log_device(C2_AGGREGATES, C2_EVENTS, C2_SEED, dev)
    out = torch.zeros((C2_AGGREGATES, 64), dtype=torch.uint8, device=dev)
    with ReplayEngine(device=local_rank) as e2:
        e2.load_csr(so, ev, None, out)
        steps = 150  # ~0.7 ms each: a > 100 ms timed region
        dt, st, times_ms = time_folds(e2, torch, dev, S.ALGO_AUTO, steps, 5)
        auto_states = out.clone()
        e2.prepare(S.ALGO_TILED)
        e2.synchronize()
        layout = e2.layout_info()
        dt_t, st_t, times_t = time_folds(e2, torch, dev, S.ALGO_TILED, steps, 5)
        return {
            "config": {"workload": f"C2: {C2_AGGREGATES} aggregates x {C2_EVENTS} events, 16 B events, 64 B state, single GPU, log resident in HBM",
                       "algo": algo_name(S, st.last_algo), "wave_tasks": st.n_tasks},
            "metric": "events/sec replayed",
            "value": C2_AGGREGATES * C2_EVENTS * steps / dt,
            "unit": "events/s",
            "steps": steps,
            "ms_per_step": dt / steps * 1e3,
            "roofline": roofline_of(S, st, times_ms),
            "tile_major": {"algo": algo_name(S, st_t.last_algo), "kernel": kernel_name(S, st_t.last_algo), "ms_per_step": dt_t / steps * 1e3,
                           "kernel_ms_min_median_max": [float(np.min(times_t)), float(np.median(times_t)), float(np.max(times_t))],
                           "frac": st_t.algorithmic_bytes / (float(np.mean(times_t)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "traffic": pmc_traffic(kernel_name(S, st_t.last_algo), st_t.algorithmic_bytes),
                           "one_shot": {"index_build_ms": layout.index_build_ms, "relayout_ms": layout.relayout_ms, "tile_major_copy_bytes": layout.tiled_bytes},
                           "states_equal_primary": bool(torch.equal(auto_states, out))},
        }


def run_c5(args):
    """BASELINE config C5: streaming micro-batches onto a GPU-resident state store.

    Population: A aggregates resident in HBM (default 10 M x 64 B), first recovered by a full fold.  A step = one micro-batch of B
    events (default 100 000 = 100 ms of a 1 M events/s stream) whose aggregate ids are Zipf-popular, in topic order, handed
    over as host buffers: pinned double-buffered staging + H2D, device group-by (stable radix sort + head scan), fold onto the
    resident state (K3), nothing waits for the device in between.  Every S batches (default 30 = 3 s,
    kafka.streams.commit-interval-ms = 3000: modules/common/src/main/resources/reference.conf:19) the state-topic delta is
    published: delta kernel -> filtered GPU encoder -> D2H -> Kafka record batches (the incremental KTable snapshot).
    `value` = events/s over the K pipelined batches INCLUDING the snapshots (barrier + sync on both sides); latency
    percentiles come from a second pass that synchronises after every batch.  The resident state after the run is checked
    byte for byte against the CPU oracle replaying the same batches."""
    import numpy as np
    import torch

    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.dist import id_table_utf16
    from surge_amd.log import batch_groups
    from surge_amd.replay import ReplayEngine
    from surge_amd.snapshot import BulkSnapshotPublisher

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the replay engine has no CPU fallback")
    dev = torch.device("cuda:0")
    A = args.aggregates or N_AGGREGATES
    B, every = args.batch_events, args.snapshot_every
    K = args.steps if args.steps != 20 else 600  # the generic default (20) means "not given": C5 is 600 batches = 60 s of stream
    W = max(args.warmup, 3)
    so, ev = synth.fixed_log_device(A, 16, 5, dev, mix=synth.C1_MIX)  # initial recovery: a short uniform log
    eng = ReplayEngine()
    eng.load_csr(so, ev)
    eng.fold()
    eng.synchronize()
    recovered = eng.snapshot()
    ids = torch.arange(A, dtype=torch.int64, device=dev)
    u16, o16 = id_table_utf16(ids)
    pub = BulkSnapshotPublisher(eng, None, N_PARTITIONS, tables=(u16.to(torch.uint8), o16.clone(), u16, o16), device_framing=not args.host_framing)
    pub.publish()  # the full snapshot after recovery: the baseline of the deltas
    snap_parts = []

    rng = np.random.default_rng(7)
    cdf = synth.zipf_cdf(4096)

    def make_batch(b):  # Zipf-popular aggregate ids (rank -> id = rank * 2654435761 mod A), events in topic order
        ranks = np.searchsorted(cdf, rng.random(B)).astype(np.int64) * (A // 4096) + rng.integers(0, max(A // 4096, 1), B)
        agg_idx = (ranks * 2654435761) % A
        words = synth.event_words(np.arange(B, dtype=np.int64) + b * B, agg_idx, np.arange(B, dtype=np.int64), 11, synth.C1_MIX)
        return agg_idx, synth.to_event_records(words)

    batches = [make_batch(b) for b in range(W + 2 * K)]
    dbatches = None
    if args.device_batches:
        dbatches = [(torch.from_numpy(a).to(dev), torch.from_numpy(e.view(np.int64).reshape(-1, 2)).to(dev)) for a, e in batches]
        torch.cuda.synchronize(dev)

    def submit(b):
        if dbatches is not None:
            eng.append_events(*dbatches[b])
        else:
            eng.append_events(*batches[b])

    for b in range(W):
        submit(b)
    eng.synchronize()
    # ---- pass 1, the metric: K batches back to back, snapshots included, one sync at the end -----------------------------
    snap_ms, touched, snap_bytes = [], [], []
    eng.stats_reset()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(K):
        submit(W + i)
        if every > 0 and (i + 1) % every == 0:
            # Synchronous: the record batches are framed on the device (DeviceFramer), the host adds the batches' CRCs — a
            # few milliseconds, nothing worth a worker thread.  (--host-framing: the C++ writer on up to 16 host threads,
            # the publish of rounds 2 and 3 until the framer existed: 48-52 ms per snapshot.)
            ts = time.perf_counter()
            out = pub.publish()
            snap_ms.append((time.perf_counter() - ts) * 1e3)
            snap_parts.append({k: v for k, v in pub.timings.items() if k.endswith("_ms")})
            touched.append(int(pub.timings["values"] + pub.timings["tombstones"]))
            snap_bytes.append(sum(len(x) for x in out.values()))
    eng.synchronize()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kern = eng.fold_times_ms()
    # ---- pass 2: per-batch latency (submit -> states resident), a sync after every batch ----------------------------------
    lat = []
    for i in range(K):
        t1 = time.perf_counter()
        submit(W + K + i)
        eng.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat = np.array(lat)
    ingest_only = B * K / float(lat.sum() / 1e3)
    # ---- parity + cpu_baseline: the CPU restatement replays the same batches onto the same recovered states ---------------
    cores, logical, quota = effective_cpus()
    cpu_baseline = None
    if not args.no_cpu_baseline and args.parity != "none":
        state = recovered
        cpu_s, cpu_events = 0.0, 0
        for b in range(W + 2 * K):
            agg_idx, events = batches[b]
            tg = time.perf_counter()
            group_agg, group_off, sorted_ev = batch_groups(agg_idx, events)
            sub = oracle.fold_csr(group_off, sorted_ev, state[group_agg])
            state[group_agg] = sub
            if b >= W and cpu_events < 100 * B:  # the timed sample: the first 100 timed batches, one host thread
                cpu_s += time.perf_counter() - tg
                cpu_events += B
        got = eng.snapshot()
        cpu_baseline = {"value": cpu_events / cpu_s, "unit": "events/s", "cores": 1, "kind": "port",
                        "sample": f"{cpu_events // B} of the same micro-batches: numpy stable group-by + the C restatement of the fold onto a host "
                                  f"copy of the state store, one host thread ({logical} logical CPUs visible, cgroup CPU quota "
                                  f"{'none' if quota is None else round(quota, 2)})",
                        "gpu_matches_cpu_full_run": bool(got.tobytes() == state.tobytes()),
                        "batches_checked": W + 2 * K}
    groups = float(np.mean([np.unique(batches[W + i][0]).size for i in range(min(K, 50))]))
    alg = 16 * B + 8 * (groups + 1) + 128 * groups  # SURVEY §8d with r = 1: events + offsets + state read and written
    kernel_ms = float(np.mean(kern)) if len(kern) else 0.0
    achieved = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    result = {
        "metric": "events/sec replayed",
        "value": B * K / elapsed,
        "unit": "events/s",
        "n_gpus": 1,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32/int64 adds + bit-copied f64",
        "data": "synthetic (counter-hash events, Zipf-popular aggregate ids; surge_amd/synth.py), generated on the host",
        "config": {
            "workload": f"C5: {A} resident aggregates, micro-batches of {B} events in topic order "
                        f"({'device-resident' if args.device_batches else 'host buffers: pinned double-buffered staging + H2D'}), device group-by + "
                        f"fold onto the resident state, state-topic delta every {every} batches (commit interval 3 s at 1 M events/s)",
            "aggregates": A, "batch_events": B, "snapshot_every": every,
            "target_ingest_events_per_sec": 1_000_000,
            "pipelined_ingest_events_per_sec_excluding_snapshots": B * K / max(elapsed - sum(snap_ms) / 1e3, 1e-9),
            "ingest_only_events_per_sec_synced_per_batch": ingest_only,
            "batch_latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
            "snapshot_ms": {"mean": float(np.mean(snap_ms)) if snap_ms else None, "max": float(np.max(snap_ms)) if snap_ms else None, "n": len(snap_ms)},
            "snapshot_framing": "host (surge_snapshot_writer, C++ threads)" if args.host_framing else "device (surge_device_framer) + host CRC-32C",
            "snapshot_parts_ms_mean": {k: float(np.mean([x[k] for x in snap_parts])) for k in (snap_parts[0] if snap_parts else {})},
            "snapshot_published_aggregates_mean": float(np.mean(touched)) if touched else None,
            "snapshot_record_batch_bytes_mean": float(np.mean(snap_bytes)) if snap_bytes else None,
            "groups_per_batch_mean": groups,
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": kernel_name(S, S.ALGO_FLAT), "kernel_ms": kernel_ms, "algorithmic_bytes": alg, "timed_launches": int(len(kern)),
                     "note": "latency-bound, not bandwidth-bound: a micro-batch is ~2 MB of events and ~8 MB of touched states; the fold kernel "
                             "is a fraction of the batch (the group-by's launches are the rest)"},
        "cpu_baseline": cpu_baseline,
    }
    pub.close()
    eng.close()
    return result


def run_v2(args):
    """ABI v2 (models the seven named v1 fields cannot express): an accumulating f64 ledger — IEEE double
    ADD / SUB strictly in event order, a running Math.max, an Int transaction count, event_count — over a Zipf(1..4096)
    log (default 2 M aggregates), one lane per aggregate.  Four timed variants of the SAME device code: compiled for the
    schema by hiprtc at create time (what a host gets) or as the generic interpreter (SURGE_REPLAY_RTC=0), each over the
    bound CSR log and over the tile-major copy.  `value` = the specialised kernels over the tile-major copy; every
    variant's states are compared with the first one's, and that one with the sequential CPU oracle on the whole log."""
    import numpy as np
    import torch

    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.replay import ReplayEngine
    from surge_amd.schema import CLS_CREATE, CLS_REQUIRE, OP_ADD, OP_MAX, OP_SET, OP_SUB, SLOT_F64, SLOT_I32, SRC_ONE, SRC_PAYLOAD, Slot, SlotAlgebra

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the replay engine has no CPU fallback")
    dev = torch.device("cuda:0")
    ledger = SlotAlgebra(
        slots=(Slot("balance", SLOT_F64, SRC_PAYLOAD), Slot("largest", SLOT_F64, SRC_PAYLOAD, default=float("-inf")), Slot("n", SLOT_I32, SRC_ONE)),
        types=((CLS_CREATE, {"balance": OP_SET}), (CLS_REQUIRE, {"balance": OP_ADD, "largest": OP_MAX, "n": OP_ADD}),
               (CLS_REQUIRE, {"balance": OP_SUB, "largest": OP_MAX, "n": OP_ADD})), count_events=True)
    n = args.aggregates or 2_000_000
    lens = synth.zipf_lengths(torch.arange(n, dtype=torch.int64, device=dev), ZIPF_SEED)
    so = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=so[1:])
    E = int(so[-1])
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ty = torch.randint(0, 100, (E,), device=dev, generator=g)
    ty = torch.where(ty < 3, 0, torch.where(ty < 55, 1, 2)).to(torch.int64)  # 3 % Opened, 52 % Credited, 45 % Debited
    val = (torch.rand(E, device=dev, generator=g, dtype=torch.float64) * 1e6).view(torch.int64)
    ev = torch.stack((ty | (torch.arange(E, device=dev) % 1000 + 1) << 32, val), dim=1)
    del ty, val
    K, W = args.steps, args.warmup
    variants, first_states, info_spec, one_shot = {}, None, None, None
    for build in ("specialised", "interpreter"):
        os.environ["SURGE_REPLAY_RTC"] = "1" if build == "specialised" else "0"
        with ReplayEngine(ledger) as eng:
            info = eng.kernel_info()
            if build == "specialised":
                info_spec = info
                if not info["specialised"]:
                    raise SystemExit("the schema-specialised kernels are not available: " + info["detail"])
            out = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
            eng.load_csr(so, ev, None, out)
            for algo, label in ((S.ALGO_TILED, "tiled"), (S.ALGO_SLOTS, "csr")):
                if algo == S.ALGO_TILED:
                    eng.prepare(algo)
                    lay = eng.layout_info()
                    if one_shot is None:
                        one_shot = {"index_build_ms": lay.index_build_ms, "relayout_ms": lay.relayout_ms, "tile_major_copy_bytes": lay.tiled_bytes,
                                    "padding_events": lay.padding_events, "schema_compile_ms": info["compile_ms"]}
                dt, st, times = time_folds(eng, torch, dev, algo, K, W)
                states = out.clone()
                if first_states is None:
                    first_states = states
                kms = float(np.mean(times))
                variants[f"{build}/{label}"] = {
                    "events_per_sec": E * K / dt, "kernel_ms": kms, "kernel_ms_min_median_max": [float(np.min(times)), float(np.median(times)), float(np.max(times))],
                    "frac": st.algorithmic_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "states_equal_first_variant": bool(torch.equal(states, first_states))}
                if build == "specialised" and label == "tiled":
                    primary = (dt, st, times)
    os.environ.pop("SURGE_REPLAY_RTC", None)
    dt, st, times = primary
    kms = float(np.mean(times))
    cpu_baseline = None
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        so_h, ev_h = so.cpu().numpy(), ev.cpu().numpy().view(S.EVENT_DTYPE).reshape(-1)
        exp = oracle.fold_csr_v2(so_h, ev_h, ledger)
        cpu_s = time.perf_counter() - t0
        parity = first_states.cpu().numpy().tobytes() == exp.tobytes()
        cpu_baseline = {"value": E / cpu_s, "unit": "events/s", "cores": 1, "kind": "port",
                        "sample": f"the whole log ({E} events, incl. the D2H copy), single-threaded C slot interpreter (oracle_fold_csr_v2)",
                        "gpu_matches_cpu_full_log": bool(parity)}
    return {
        "metric": "events/sec replayed", "value": E * K / dt, "unit": "events/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 add / sub / Math.max in event order + int32 add",
        "data": "synthetic (Zipf(1..4096) ledger log generated on the device)",
        "config": {"workload": f"ABI v2 ledger schema (f64 balance += / -= amount in event order, f64 largest = Math.max, i32 transactions, event_count) "
                               f"over {n} aggregates, Zipf(1..4096) events each, one lane per aggregate; log resident in HBM",
                   "aggregates": n, "events": E, "kernels": info_spec["detail"], "variants": variants},
        "roofline": {"bound": "hbm", "achieved": st.algorithmic_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": st.algorithmic_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": pmc_traffic("surge_slots_tiled2", st.algorithmic_bytes)[0], "kernel": "surge_slots_tiled2",
                     "kernel_note": "fold_slots_device.h compiled for the schema by hiprtc at surge_replay_create_v2",
                     "kernel_ms": kms, "kernel_ms_min_median_max": [float(np.min(times)), float(np.median(times)), float(np.max(times))],
                     "algorithmic_bytes": st.algorithmic_bytes, "timed_launches": int(len(times))},
        "one_shot": one_shot,
        "cpu_baseline": cpu_baseline,
    }


def init_ranks(args, torch):
    """One process per GPU: starts the ranks when bench.py was called without a launcher, joins the process group.
    Returns (world, rank, local_rank, dev, ctl, dist, rehearsal)."""
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start one rank per GPU ourselves (what the driver's torchrun line does)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the replay engine has no CPU fallback")
    # Rehearsal (SURGE_BENCH_REHEARSAL=1): every rank on cuda:0, gloo control plane on CPU tensors — exercises the whole
    # N > 1 code path on a one-GPU box (with SURGE_RCCL_LIBRARY = tests/rccl_stub for the data exchange, since RCCL refuses
    # two ranks on one device).  The line it prints is labelled; its throughput means nothing.
    rehearsal = world > 1 and os.environ.get("SURGE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = torch.device("cpu") if rehearsal else dev  # where the control-plane tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # control plane: barriers, the timing reductions, the communicator id
    return world, rank, local_rank, dev, ctl, dist, rehearsal


E2E_SEED = 11


def e2e_events(np, synth, agg, j):
    """The Counter event aggregate ``agg`` publishes as its j-th (1-based): (type 0 increment / 1 decrement / 2 no-op,
    argument 0..999), a pure function of (agg, j) — the topic generator and the parity check both call it."""
    idx = agg.astype(np.int64) * 8192 + j
    return (synth._h(E2E_SEED, idx, 21) % 3).astype(np.int32), (synth._h(E2E_SEED, idx, 22) % 1000).astype(np.int32)


class CounterTopic:
    """The e2e topic's default model: Counter fixture events (TestBoundedContext.scala:42-49,122-124) on the C3 / C4 population."""
    name = "counter"
    key_bytes = 13
    headers = ()

    def __init__(self, np, synth, S):
        from fixture_models import CT_DEC, CT_INC, CT_NOOP, CounterBusinessLogic

        self.np, self.synth, self.S = np, synth, S
        bl = CounterBusinessLogic()
        self.model, self.fmt = bl.command_model(), bl.event_write_formatting()
        self.types = np.array([CT_INC, CT_DEC, CT_NOOP], np.int32)
        self.data = "Counter fixture events as play-json text"
        self.population = "the C3 / C4 population: {A} aggregates (acct-%08d, Zipf(1..4096) counts, seed {seed}), the first <= {cap} events of each"

    def partitions(self, A, P, eng, dev, torch):
        from surge_amd.dist import partitions_of_ids

        ids_all = torch.arange(A, dtype=torch.int64, device=dev)
        return partitions_of_ids(ids_all, P, eng).cpu().numpy().astype(self.np.int32)

    def counts(self, ids, cap):
        return self.np.minimum(self.synth.zipf_lengths(ids, ZIPF_SEED), cap).astype(self.np.int64)

    def records(self, topic_gen, a, j):
        ty, arg = e2e_events(self.np, self.synth, a, j)
        return topic_gen.counter_records(a, ty, arg, j)

    def check_sample(self, a, j, i, key, value):
        from fixture_models import CountDecremented, CountIncremented, NoOpEvent

        ty, arg = e2e_events(self.np, self.synth, a[i:i + 1], j[i:i + 1])
        agg = f"acct-{a[i]:08d}"
        e = [CountIncremented(agg, int(arg[0]), int(j[i])), CountDecremented(agg, int(arg[0]), int(j[i])), NoOpEvent(agg, int(j[i]))][ty[0]]
        m = self.fmt.write_event(e)
        assert key == m.key.encode() and value == m.value

    def ids_of_keys(self, kb, koff, n_keys):
        np = self.np
        assert np.all(np.diff(koff) == 13), "every key is an acct-%08d id"
        digits = kb[: 13 * n_keys].reshape(n_keys, 13)[:, 5:].astype(np.int64) - 48
        return (digits * (10 ** np.arange(7, -1, -1, dtype=np.int64))).sum(axis=1)

    def source_events(self, ev_agg, ev_j):
        ty, arg = e2e_events(self.np, self.synth, ev_agg, ev_j)
        src = self.np.zeros(ev_agg.shape[0], dtype=self.S.EVENT_DTYPE)
        src["type"] = self.types[ty]
        src["seq"] = ev_j
        src["raw"] = arg.astype(self.np.uint32).astype(self.np.uint64)
        return src


class MixedTopic:
    """--e2e-topic mixed: what the Counter topic leaves out — the surge-docs BankAccount model
    (BankAccountSurgeModel.scala:26-32, BankAccountCommandModel.scala:81-86): record keys are the account's UUID alone (36
    bytes, no ':' and no sequence number), the balances are Doubles as play-json text (parsed to the bit on the device),
    every record carries two headers the decoder has to step over, and every 128th account publishes 64..256 events where
    the others publish <= --events-cap — a fetch of a late round holds the same few thousand accounts dozens of times."""
    name = "mixed"
    key_bytes = 36
    headers = (("traceparent", b"00-0af7651916cd43dd8448eb211c80319c-b7ad6b7169203331-01"), ("content-type", b"application/json"))
    LONG_EVERY, LONG_MIN, LONG_SPAN = 128, 64, 193

    def __init__(self, np, synth, S):
        from fixture_models import BA_CREATED, BA_UPDATED, BankAccountCommandModel, BankAccountEventFormat

        self.np, self.synth, self.S = np, synth, S
        self.model, self.fmt = BankAccountCommandModel(), BankAccountEventFormat()
        self.types = (BA_CREATED, BA_UPDATED)
        self.data = "surge-docs BankAccount events (UUID keys, Double balances) as play-json text, two record headers each"
        self.population = ("{A} accounts (UUID ids), Zipf(1..4096) counts (seed {seed}) cut to <= {cap} events each except every 128th account, which publishes 64..256; "
                           "every record with a traceparent and a content-type header")

    def cents(self, a, j):
        return 1 + (self.synth._h(E2E_SEED, a.astype(self.np.int64) * 8192 + j, 22) % 999_999_998).astype(self.np.int64)

    def partitions(self, A, P, eng, dev, torch):
        import ctypes

        import topic_gen
        from surge_amd import _native

        np = self.np
        out = np.zeros(A, np.int32)
        step = 1 << 20
        for lo in range(0, A, step):  # partitionForKey of the UUID strings (KafkaPartitioner.scala:8), through the C ABI's host entry point
            ids = np.arange(lo, min(A, lo + step), dtype=np.int64)
            k, ko, _, _ = topic_gen.bank_records(ids, np.full(ids.shape[0], 2, np.int32), np.ones(ids.shape[0], np.int64))
            utf16 = k.astype(np.uint16)
            part = np.zeros(ids.shape[0], np.int32)
            rc = _native.load().surge_replay_partition_hash(utf16.ctypes.data_as(ctypes.c_void_p), np.ascontiguousarray(ko).ctypes.data_as(ctypes.c_void_p), ids.shape[0], P,
                                                            part.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0
            out[lo:lo + ids.shape[0]] = part
        return out

    def counts(self, ids, cap):
        np = self.np
        c = np.minimum(self.synth.zipf_lengths(ids, ZIPF_SEED), cap).astype(np.int64)
        long_rows = self.synth._h(E2E_SEED, ids, 24) % self.LONG_EVERY == 0
        return np.where(long_rows, self.LONG_MIN + (self.synth._h(E2E_SEED, ids, 25) % self.LONG_SPAN).astype(np.int64), c)

    def records(self, topic_gen, a, j):
        return topic_gen.bank_records(a, j, self.cents(a, j))

    def check_sample(self, a, j, i, key, value):
        import uuid

        from fixture_models import BankAccountCreated, BankAccountUpdated

        u = uuid.UUID(key.decode())
        assert str(u) == key.decode() and int(key[-12:], 16) == int(a[i])
        amount = int(self.cents(a[i:i + 1], j[i:i + 1])[0]) / 100
        e = BankAccountCreated(u, f"Owner {int(a[i]) % 1000}", f"{int(a[i]) % 10000:04d}", amount) if j[i] == 1 else BankAccountUpdated(u, amount)
        m = self.fmt.write_event(e)
        assert key == m.key.encode() and value == m.value, (value, m.value)

    def ids_of_keys(self, kb, koff, n_keys):
        np = self.np
        assert np.all(np.diff(koff) == 36), "every key is a UUID"
        hexd = kb[: 36 * n_keys].reshape(n_keys, 36)[:, 24:].astype(np.int64)
        val = np.where(hexd >= 97, hexd - 87, hexd - 48)
        return (val << (4 * np.arange(11, -1, -1, dtype=np.int64))).sum(axis=1)

    def source_events(self, ev_agg, ev_j):
        np = self.np
        src = np.zeros(ev_agg.shape[0], dtype=self.S.EVENT_DTYPE)
        src["type"] = np.where(ev_j == 1, self.types[0], self.types[1]).astype(np.int32)
        src["seq"] = 0  # (these events carry no sequence number: BankAccountCommandModel.encode_event)
        src["raw"] = (self.cents(ev_agg, ev_j) / 100.0).view(np.uint64)
        return src


def run_e2e(args):
    """From the bytes Kafka hands over to recovered states, on the population BASELINE.json's target is quoted on
    (SURVEY §8f N1 in front of R2; the recovery SurgeStateStoreConsumer.scala:57-76 performs record by record).

    Topic: the 10,000,000 aggregates of config C3 / C4 (ids ``acct-%08d``, Zipf(1..4096) event counts, seed 3), every
    aggregate publishing the first min(count, --events-cap) of its Counter events (play-json text as the reference writes
    it, TestBoundedContext.scala:42-49,122-124; keys ``<id>:<seq>``), interleaved in rounds — round j holds the j-th event
    of every aggregate that has one, in a fixed pseudo-random order, so a fetch of a million records touches a million
    different aggregates: the key table and the resident state see no locality at all — over 64 partitions by the
    reference's partitioner (KafkaPartitioner.scala:8,38-42), lz4-compressed like the reference's producer
    (reference.conf:112-115) and, by default, SHAPED like its publisher's output: per partition one transaction per flush of
    --txn-flush-events records (KafkaProducerActorImpl.scala:421-453) — data batches closed by the flush or at 16 KiB, then a
    COMMIT control batch; every --abort-every-th flush first fails (records + ABORT marker) and is retried; on a quarter of
    the partitions a response's last marker arrives a fetch late — written by the independent test-side producer
    tests/native/wire_writer.c (--writer product: the product's own record-batch writer, non-transactional, every batch
    filled to 16 KiB: round 4's topic; --txn-flush-events 0: that layout from the independent writer).  With N GPUs rank
    r consumes the partitions p % N == r (PartitionAssignments.scala:51-63) and the final snapshot is all-gathered through
    the C ABI.

    Path, per rank: a fetch response = the next ~--batch-events records of the rank's partitions -> host framing (headers,
    CRC-32C, read_committed) per partition on --framing-threads threads, one fetch ahead -> ONE device push per fetch
    (surge_device_decoder_push_parts_async: copy, LZ4 blocks, records, JSON -> 16-byte events; four in flight) ->
    key interning -> device group-by + fold onto the resident state (K3).  A step = one fetch; `value` = events/s from the
    completed fold of the last warm-up fetch to the completed fold of the last fetch, over all ranks, framing included.
    The states after the run are compared, aggregate by aggregate, with the oracle's fold of the SOURCE events (the
    (agg, j) -> event function the generator wrote the topic from), not of anything the device decoded."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import topic_gen
    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.dist import shard_of_partition
    from surge_amd.ingest import DeviceDecoder, EventsTopicIngest, PartitionedFramedFetches, PushPipeline
    from surge_amd.replay import ReplayEngine
    from surge_amd.snapshot import RecordBatchWriter

    world, rank, local_rank, dev, ctl, dist, rehearsal = init_ranks(args, torch)
    mixed = args.e2e_topic == "mixed"
    A = args.aggregates or (4_000_000 if mixed else N_AGGREGATES)
    cap = args.events_cap
    P = N_PARTITIONS
    n_fetch = args.batch_events if args.batch_events != 100_000 else 1_000_000
    W = min(args.warmup, 2)
    depth = 1 if args.serial_framing else int(os.environ.get("SURGE_BENCH_DEPTH", "4"))
    one_thread = args.serial_framing or not args.two_thread_consumer
    topic = (MixedTopic if mixed else CounterTopic)(np, synth, S)
    model = topic.model
    tmpl = model.event_json_template()
    t_gen = time.perf_counter()

    eng = ReplayEngine(model.event_algebra(), device=local_rank)
    # ---- this rank's aggregates: its partitions' -------------------------------------------------------------------------
    part_all = topic.partitions(A, P, eng, dev, torch)
    mine = shard_of_partition(part_all, world) == rank
    my_ids = np.flatnonzero(mine).astype(np.int64)
    my_part = part_all[mine].astype(np.int32)
    del part_all, mine
    counts = topic.counts(my_ids, cap)
    order = np.argsort(synth._h(E2E_SEED, my_ids, 23), kind="stable")  # the order aggregates take turns in, every round
    my_ids, my_part, counts = my_ids[order], my_part[order], counts[order]
    n_total = int(counts.sum())
    if args.steps != 20:  # an explicit --steps: that many timed fetches
        n_total = min(n_total, (W + args.steps) * n_fetch)
    # ---- the topic, fetch by fetch: (agg, j) stream -> record text -> record batches per partition ----------------------
    fetches, n_pub = [], 0
    wire_bytes = 0
    incl = np.zeros(my_ids.shape[0], np.int64)  # events of each aggregate that made it into the topic
    sample_checked = False
    independent = args.writer == "independent"
    K_flush = args.txn_flush_events if independent else 0
    if topic.headers and not independent:
        raise SystemExit("--e2e-topic mixed writes record headers: only the independent writer (tests/native/wire_writer.c) does")
    header_bytes = topic_gen.set_record_headers(topic.headers) if independent else 0
    with (topic_gen.WireTopic(P, K_flush, 16384, args.codec, args.abort_every if K_flush else 0, args.hold_markers if K_flush else 0) if independent
          else RecordBatchWriter(P, 0, 16384, args.codec)) as writer:
        pend_a, pend_j, pend_p, pend_n = [], [], [], 0

        def flush(final=False):
            nonlocal pend_a, pend_j, pend_p, pend_n, wire_bytes, sample_checked
            while pend_n >= n_fetch or (final and pend_n > 0):
                a, j, p = np.concatenate(pend_a), np.concatenate(pend_j), np.concatenate(pend_p)
                take = min(n_fetch, a.shape[0])
                pend_a, pend_j, pend_p, pend_n = [a[take:]], [j[take:]], [p[take:]], a.shape[0] - take
                a, j, p = a[:take], j[:take], p[:take]
                k, ko, v, vo = topic.records(topic_gen, a, j)
                if not sample_checked:  # the generator's text IS what the fixture's event writer writes
                    for i in range(0, take, max(1, take // 200)):
                        topic.check_sample(a, j, i, bytes(k[ko[i]:ko[i + 1]]), bytes(v[vo[i]:vo[i + 1]]))
                    sample_checked = True
                # (the topic's last response holds no marker back: what an earlier response held back arrives with it)
                last = final and pend_n == 0 or n_pub >= n_total and pend_n == 0
                parts = writer.fetch(p, k, ko, v, vo, last=last) if independent else topic_gen.frame_partitions(writer, p, k, ko, v, vo)
                wire_bytes += sum(len(x) for x in parts if x)
                fetches.append((parts, take))

        for j in range(1, int(counts.max()) + 1 if counts.shape[0] else 1):
            if n_pub >= n_total:
                break
            sel = np.flatnonzero(counts >= j)
            if n_pub + sel.shape[0] > n_total:
                sel = sel[: n_total - n_pub]
            if not sel.shape[0]:
                break
            incl[sel] += 1
            n_pub += sel.shape[0]
            pend_a.append(my_ids[sel]); pend_j.append(np.full(sel.shape[0], j, np.int32)); pend_p.append(my_part[sel]); pend_n += sel.shape[0]
            flush()
        flush(final=True)
        if independent and K_flush and args.hold_markers:
            # the markers the last fetch response held back: a response of control batches only (the transactions they close become visible with it)
            parts = writer.fetch(np.zeros(0, np.int32), None, None, None, None)
            if any(parts):
                wire_bytes += sum(len(x) for x in parts if x)
                fetches.append((parts, 0))
        topic_counts = writer.counts if independent else None
    if header_bytes:
        topic_gen.set_record_headers(())  # (a process-wide setting of the test writer)
    gen_s = time.perf_counter() - t_gen
    K = len(fetches) - W
    if K < 1:
        raise SystemExit(f"--workload e2e: the topic holds {len(fetches)} fetch(es): nothing to time behind {W} warm-up fetch(es)")

    if getattr(args, "host_only", False):
        return run_e2e_host_only(args, torch, np, fetches, W, world, rank, dist, ctl, wire_bytes, n_pub, gen_s, topic_counts, PartitionedFramedFetches, P)
    # ---- the run --------------------------------------------------------------------------------------------------------
    marks, dev_ms, keys_at, push_ms = [], [], [], []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    # host threads: the CPUs this process may use are shared by the node's ranks — each rank frames on its share minus the
    # consumer thread (one rank: minus the consumer, the framing driver and the Python main thread's neighbours)
    args.framing_threads = max(1, min(args.framing_threads, int(effective_cpus()[0]) // world - (3 if world == 1 else 1)))
    bound_log = bool(getattr(args, "bound_log", False))
    cpu_t0 = [None]
    consumer_cpu, push_cpu = [], []  # the consumer thread's CPU seconds per fetch: (finish, fold) and push_async
    by_copy = bool(getattr(args, "framing_by_copy", False))
    with PartitionedFramedFetches((f for f, _ in fetches), P, threads=args.framing_threads, hold=depth, overlap=not args.serial_framing, device_crc=not by_copy,
                                  in_place=not by_copy) as framed, \
            DeviceDecoder(tmpl, device=local_rank) as d:
        # capacity hints (a recovery knows roughly how many aggregates the store held: its last snapshot): the resident state
        # and the decoder's key table are sized for this rank's aggregates up front instead of growing step by step
        # (--no-capacity-hint: every growth step allocates, copies and frees device memory — tens of ms each at this size)
        hint = 0 if args.no_capacity_hint else int(my_ids.shape[0])
        eng.load_csr(np.zeros(hint + 1, np.int64), np.zeros(0, dtype=S.EVENT_DTYPE))
        eng.fold()
        if hint:
            d.reserve(hint, topic.key_bytes * hint)
        n_agg = hint

        def finish_one(wait):
            nonlocal n_agg
            t1 = time.perf_counter()
            c1 = time.thread_time()
            d.finish(wait=(wait is True or wait == "finish"))
            ta = time.perf_counter()
            ca = time.thread_time()
            n_keys = d.n_keys
            if n_keys > n_agg:
                eng.grow(max(n_keys, min(2 * n_agg, my_ids.shape[0])))  # (grow in big steps: a grow copies the resident state)
                n_agg = eng.n_agg
            tb = time.perf_counter()
            if bound_log:
                d.stage_into(eng)  # no fold: the topic's end packs what was staged (below)
                if wait is True:
                    eng.synchronize()
            else:
                d.fold_into(eng, wait=(wait is True))
            t2 = time.perf_counter()
            consumer_cpu.append((ca - c1, time.thread_time() - ca))
            if len(marks) == W - 1:
                cpu_t0[0] = (time.process_time(), framed.cpu_seconds(), thread_cpu_seconds())  # host CPU seconds (every thread of this process; the framing threads' own) from the end of the warm-up on
            if os.environ.get("SURGE_BENCH_TRACE"):
                print(f"[bench] fetch {len(marks)}: finish {(ta - t1) * 1e3:.2f} grow {(tb - ta) * 1e3:.2f} fold {(t2 - tb) * 1e3:.2f} ms, keys {n_keys}", file=sys.stderr)
            marks.append(t2)
            dev_ms.append((t2 - t1) * 1e3)
            keys_at.append(n_keys)

        if one_thread:
            # one host thread enqueues stage 1 of fetch i + depth, then finishes fetch i and folds it
            waits = getattr(args, "consumer_waits", "both")
            pending = 0
            fetch_iter = iter(framed)
            # (the engine stays on the decoder's stream: with the three low-priority push streams that is the arrangement of DESIGN section 7;
            # --consumer-waits: both = a host wait behind interning and behind the fold, finish = behind interning only — the fold is handed
            # over by an event —, none = only the wait in the middle of interning)
            while True:
                # the oldest push is finished BEFORE the next fetch is asked for: asking tells the framer that the oldest fetch's
                # slab may be framed into again, and a push reads its bytes until it is finished
                if pending == depth:
                    finish_one(True if waits == "both" else waits)
                    pending -= 1
                parts = next(fetch_iter, None)
                if parts is None:
                    break
                tp, cp = time.perf_counter(), time.thread_time()
                d.push_async(parts)
                push_ms.append((time.perf_counter() - tp) * 1e3)
                push_cpu.append(time.thread_time() - cp)
                pending += 1
            while pending:
                finish_one(True if waits == "both" else waits)
                pending -= 1
            eng.synchronize()
        else:
            # three host threads: the framer's driver, the pipeline's worker (stage 1 of up to `depth` fetches ahead: section
            # tables, staging, launches) and this one (stage 2 + fold of the oldest push, no host wait behind either: the
            # decoder's stream and the engine's — a stream of its own here — are ordered by events)
            def push(parts):
                d.push_async(parts)
                return True

            with eng.on_own_stream(), PushPipeline(framed, push, depth) as pipe:
                for _ in pipe:
                    finish_one(False)
                    pipe.done()
                eng.synchronize()
            push_ms = [x * 1e3 for x in pipe.push_seconds]
        torch.cuda.synchronize(dev)
        marks[-1] = time.perf_counter()  # (the last fetch counts as done when the device is)
        cpu_s = time.process_time() - (cpu_t0[0][0] if cpu_t0[0] is not None else 0.0)
        grp_cpu = framed.cpu_seconds()
        thr1, thr0 = thread_cpu_seconds(), (cpu_t0[0][2] if cpu_t0[0] is not None else {})
        by_thread = {}
        for tid, (name, sec) in thr1.items():
            name = "hip runtime / other native threads" if name in ("python", "python3", "pt_main_thread") else name
            by_thread[name] = by_thread.get(name, 0.0) + sec - thr0.get(tid, (name, 0.0))[1]
        if cpu_t0[0] is not None and cpu_s - sum(by_thread.values()) > 0:
            by_thread["threads that ended before the timed region did (the framing driver)"] = cpu_s - sum(by_thread.values())
        recv_cpu_s, framing_cpu_s = (grp_cpu[0] - cpu_t0[0][1][0], grp_cpu[1] - cpu_t0[0][1][1]) if cpu_t0[0] is not None else grp_cpu
        torch.cuda.synchronize(dev)
        t_begin = marks[W - 1] if W > 0 else t_start
        elapsed_local = marks[-1] - t_begin
        n_keys = keys_at[-1]
        packed = None
        if bound_log:
            # the topic's end: everything staged -> ONE bound CSR log -> ONE fold (what AUTO picks for it); then the re-fold rate
            n_staged = eng.staged
            tp0 = time.perf_counter()
            eng.pack_staged(max(n_keys, hint))
            eng.synchronize()
            tp1 = time.perf_counter()
            eng.fold()
            eng.synchronize()
            tp2 = time.perf_counter()
            lay = eng.layout_info()
            dt_r, st_r, tm_r = time_folds(eng, torch, dev, S.ALGO_AUTO, 10, 1)
            packed = {"staged_events": n_staged, "pack_ms": (tp1 - tp0) * 1e3, "first_fold_ms_incl_index": (tp2 - tp1) * 1e3, "index_build_ms": lay.index_build_ms,
                      "algo": algo_name(S, st_r.last_algo), "kernel": kernel_name(S, st_r.last_algo), "refold_kernel_ms": float(np.mean(tm_r)),
                      "refold_events_per_s": n_staged / (float(np.mean(tm_r)) * 1e-3), "refold_frac_of_8TBps": st_r.algorithmic_bytes / (float(np.mean(tm_r)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "bytes_to_states_events_per_s_incl_pack_and_fold": sum(n for _, n in fetches[W:]) / (elapsed_local + (tp2 - tp0)),
                      "log_bytes": st_r.algorithmic_bytes}
        states = eng.snapshot()[:n_keys]
        stats = d.stats()
        # the key table, as numbers (ids are acct-%08d: 13 bytes each)
        import ctypes

        lib = d._lib
        nk, nb = ctypes.c_int64(), ctypes.c_int64()
        lib.surge_device_decoder_keys(d._h, None, 0, None, ctypes.byref(nk), ctypes.byref(nb))
        kb = np.zeros(max(nb.value, 1), np.uint8)
        koff = np.zeros(nk.value + 1, np.int64)
        lib.surge_device_decoder_keys(d._h, kb.ctypes.data_as(ctypes.c_void_p), kb.shape[0], koff.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nk), ctypes.byref(nb))
        host_ms = [x * 1e3 for x in framed.framing_seconds]
        recv_ms = [x * 1e3 for x in framed.receive_seconds]
        ingest_counters = framed.counters()
    key_ids = topic.ids_of_keys(kb, koff, n_keys)
    # ---- parity: the oracle folds the SOURCE events of every aggregate, in the device's key order ------------------------
    t_par = time.perf_counter()
    sorter = np.argsort(my_ids)
    at = sorter[np.searchsorted(my_ids, key_ids, sorter=sorter)]
    assert np.array_equal(my_ids[at], key_ids) and np.unique(key_ids).shape[0] == n_keys == int((incl > 0).sum()), "the key table is not this rank's published aggregates, each once"
    cnt_k = incl[at]
    off = np.zeros(n_keys + 1, np.int64)
    np.cumsum(cnt_k, out=off[1:])
    ev_agg = np.repeat(key_ids, cnt_k)
    ev_j = (np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], cnt_k) + 1).astype(np.int32)
    src = topic.source_events(ev_agg, ev_j)
    t_or = time.perf_counter()
    if args.parity == "none":  # (A/B runs of one kernel against another: the states are not checked, and the line says so)
        exp, oracle_s, parity = None, 0.0, True
    else:
        exp = oracle.fold_csr(off, src, None, model.event_algebra(), threads=int(effective_cpus()[0]))
        oracle_s = time.perf_counter() - t_or
        parity = bool(states.tobytes() == exp.tobytes()) and int(off[-1]) == n_pub
    del src, ev_agg, ev_j
    parity_s = time.perf_counter() - t_par
    # ---- N > 1: the final snapshot to every rank ----------------------------------------------------------------------------
    exchange = None
    n_events_timed = sum(n for _, n in fetches[W:])
    elapsed = torch.tensor([elapsed_local], dtype=torch.float64, device=ctl)
    totals = torch.tensor([n_events_timed, n_keys, int(parity), wire_bytes, n_pub], dtype=torch.int64, device=ctl)
    per_rank = [n_events_timed]
    if dist is not None:
        from surge_amd.dist import NativeSnapshotGather

        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        mine_t = totals.clone()
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        allr = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        per_rank = [int(t[0].item()) for t in allr]
        parity_all = all(int(t[2].item()) == 1 for t in allr)
        gather = NativeSnapshotGather(n_keys, dev, eng)
        local = gather.make_local_buffers()[0]
        local[:n_keys] = torch.from_numpy(states.view(np.uint8).reshape(n_keys, 64)).to(dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        ta = time.perf_counter()
        gather.launch(0, local, None)
        gather.synchronize(0)
        torch.cuda.synchronize(dev)
        exchange = (time.perf_counter() - ta) * 1e3
        res = gather.result(0)
        own = torch.stack([local[:n_keys].reshape(-1).view(torch.int64).sum()]).to(ctl)
        sums = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(sums, own)
        for r in range(world):
            got = res[r, : gather.counts[r]].reshape(-1).view(torch.int64).sum()
            assert int(got.item()) == int(sums[r].item()), f"rank {rank}: gathered block of rank {r} differs from its shard"
        assert torch.equal(res[rank, :n_keys], local[:n_keys]), "all-gathered snapshot does not contain the local shard"
        exm = torch.tensor([exchange], dtype=torch.float64, device=ctl)
        dist.all_reduce(exm, op=dist.ReduceOp.MAX)
        exchange = float(exm.item())
        gathered_aggregates = int(sum(gather.counts))
    else:
        parity_all = parity
        gathered_aggregates = None
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return None
    elapsed_s = float(elapsed.item())
    total_events, total_keys, total_wire = int(totals[0].item()), int(totals[1].item()), int(totals[3].item())
    # the library's host decoder on a bounded sample of the same fetches (one thread), for scale
    t0 = time.perf_counter()
    sample_records = 0
    handles = [EventsTopicIngest() for _ in range(P)]  # (one handle per partition: read_committed state is a partition's)
    try:
        for parts, _ in fetches[W:W + 2]:
            for q, data in enumerate(parts):
                if data:
                    handles[q].feed(data)
                    sample_records += int(handles[q].drain_json(tmpl)[0].shape[0])
    finally:
        for gh in handles:
            gh.close()
    host_decoder_s = time.perf_counter() - t0
    lat = [(marks[i] - marks[i - 1]) * 1e3 for i in range(max(W, 1), len(marks))]
    if os.environ.get("SURGE_BENCH_TRACE"):
        print("[bench] ms between completed folds:", " ".join(f"{x:.2f}" for x in lat), file=sys.stderr)
        print("[bench] framing ms per fetch:", " ".join(f"{x:.2f}" for x in host_ms), file=sys.stderr)
        print("[bench] push_async host ms:", " ".join(f"{x:.2f}" for x in push_ms), file=sys.stderr)
        print("[bench] records per fetch:", " ".join(str(n) for _, n in fetches), file=sys.stderr)
        print("[bench] wire bytes per fetch:", " ".join(str(sum(len(x) for x in parts if x)) for parts, _ in fetches), file=sys.stderr)
    disc = [i for i in range(max(W, 1), len(marks)) if keys_at[i] > keys_at[i - 1] + fetches[i][1] // 2]  # fetches that mostly discover keys
    steady = [i for i in range(max(W, 1), len(marks)) if keys_at[i] == keys_at[i - 1]]
    rate = lambda idx: (sum(fetches[i][1] for i in idx) / sum(marks[i] - marks[i - 1] for i in idx)) if idx else None  # noqa: E731
    out = {
        "metric": "events/sec replayed", "value": total_events / elapsed_s, "unit": "events/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed_s / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bytes -> f64 events -> f64 moves" if mixed else "bytes -> int32 events -> int32/int64 adds", "data": f"synthetic ({topic.data} in Kafka record batches v2, written by "
                                                                   + ("the independent test-side producer tests/native/wire_writer.c — no code shared with the library that reads them)" if independent
                                                                      else "the product's record-batch writer)"),
        "config": {"workload": f"E2E: events-topic bytes -> states on " + topic.population.format(A=A, seed=ZIPF_SEED, cap=cap) + ", "
                               f"published in rounds (a fetch touches as many different aggregates as it has records), {P} partitions by "
                               f"partitionForKey, {args.codec} batches, " + (f"one transaction per publisher flush of {K_flush} records per partition (data batches closed by the flush or at 16 KiB, "
                               f"then a COMMIT control batch; every {args.abort_every}th flush aborted and retried; on partitions p % {args.hold_markers} == 1 a response's last marker "
                               f"arrives a fetch late)" if K_flush else "not transactional, every batch filled to 16 KiB") + f"; fetches of {n_fetch} records per rank; host framing (headers, CRC-32C, "
                               f"transactions) per partition on {args.framing_threads} threads one fetch ahead; one device push per fetch, {depth} in flight: LZ4 / records / "
                               f"JSON decode / key interning / group-by / fold on the GPU" + (" [REHEARSAL: every rank on cuda:0, throughput meaningless]" if rehearsal else ""),
                   "parallelism": f"partitions p % {world} == rank; no data-path collective; final snapshot all-gathered through the C ABI" if world > 1 else "one GPU",
                   "topic_model": topic.name, "record_header_bytes": header_bytes, "max_events_of_one_aggregate": int(counts.max()) if counts.shape[0] else 0,
                   "writer": args.writer, "txn_flush_events": K_flush, "topic": topic_counts,
                   "control_batches": None if topic_counts is None else topic_counts["control_batches"],
                   "aggregates": A, "events_cap": cap, "partitions": P, "fetch_records": n_fetch, "fetches": len(fetches), "pushes_in_flight": depth,
                   "consumer": ("one thread, a host wait behind interning and behind the fold" if getattr(args, "consumer_waits", "both") == "both" else
                                "one thread, a host wait behind interning; the fold handed over by an event" if getattr(args, "consumer_waits", "both") == "finish" else
                                "one thread, no host wait behind interning or the fold (event-ordered hand-over)") if one_thread else "push worker thread + finisher, event-ordered hand-over to the fold (no host wait behind either)",
                   "bound_log": packed,
                   "host_cpu_ms_per_1e6_records": cpu_s * 1e3 / max(1, n_events_timed) * 1e6,
                   "host_cpu_ms_per_1e6_records_without_the_receive_copy": (cpu_s - recv_cpu_s) * 1e3 / max(1, n_events_timed) * 1e6,
                   "framing_cpu_ms_per_1e6_records": framing_cpu_s * 1e3 / max(1, n_events_timed) * 1e6,
                   "receive_copy_cpu_ms_per_1e6_records": recv_cpu_s * 1e3 / max(1, n_events_timed) * 1e6,
                   "host_cpu_ms_per_1e6_records_by_thread": {k: round(v * 1e3 / max(1, n_events_timed) * 1e6, 3) for k, v in sorted(by_thread.items(), key=lambda kv: -kv[1]) if v > 0},
                   "consumer_cpu_ms_per_fetch": {"push_async": float(np.mean(push_cpu[W:])) * 1e3 if len(push_cpu) > W else None,
                                                 "finish (interning; one wait in the middle, one behind)": float(np.mean([c[0] for c in consumer_cpu[W:]])) * 1e3,
                                                 "fold (group-by + FLAT, one wait behind)": float(np.mean([c[1] for c in consumer_cpu[W:]])) * 1e3,
                                                 "waits": os.environ.get("SURGE_INGEST_WAIT", "poll")},
                   "receive_copy_ms_per_fetch": float(np.mean(recv_ms[W:])) if len(recv_ms) > W else None,
                   "host_cpu_note": "process CPU time (every thread: framing pool, framing driver, consumer — the consumer spins in its waits for the device) over the timed fetches of "
                                    "rank 0.  In-place framing: a fetch response is RECEIVED into the framer's page-locked slab — here one memmove per partition out of the topic's "
                                    "bytes objects, standing in for the socket read a consumer aims there (a read it performs either way): its CPU time is reported and subtracted separately",
                   "framing": "by copy, CRC-32C on the host (rounds 4 / 5)" if by_copy else "in place (no copy of the sections), CRC-32C finished on the device",
                   "framing_threads": args.framing_threads, "capacity_hint": not args.no_capacity_hint, "events_timed": total_events, "per_rank_events": per_rank, "keys_interned": total_keys,
                   "wire_bytes_per_record": total_wire / max(1, int(totals[4].item())),
                   "fetch_ms": {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)), "max": float(np.max(lat))},
                   "host_framing_ms_per_fetch": float(np.mean(host_ms[W:])), "finish_and_fold_ms_per_fetch": float(np.mean(dev_ms[W:])),
                   "push_async_host_ms_per_fetch": float(np.mean(push_ms[W:])),
                   "events_per_s_while_discovering_keys": rate(disc), "events_per_s_all_keys_known": rate(steady),
                   "decoder": stats, "ingest": ingest_counters, "generate_s": gen_s, "parity_s": parity_s,
                   "snapshot_exchange_ms": exchange, "gathered_aggregates": gathered_aggregates},
        "roofline": {"bound": "hbm", "achieved": total_wire / elapsed_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": total_wire / elapsed_s / 1e9 / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "surge_device_decoder (lz4_parse + lz4_exec + section_kernel + interning) + K3",
                     "note": "wire bytes per second against the HBM peak: this path is not bandwidth-bound — its kernels are latency- / instruction-bound (serial LZ4 "
                             "sequence walks, byte-wise JSON scans; profiles/r04_e2e_*): the kernel-level split is in the committed rocprofv3 summary, the roofline "
                             "figure of the fold this path feeds is the default workload's"},
        "cpu_baseline": {"value": sample_records / host_decoder_s, "unit": "events/s", "cores": 1, "kind": "port",
                         "sample": f"{sample_records} records of the same fetches through the library's host decoder (surge_ingest_feed + surge_ingest_drain_json), decode only — no fold",
                         "oracle_fold_events_per_s": int(off[-1]) / oracle_s if oracle_s > 0 else None,
                         "gpu_states_match_cpu_fold_of_the_source_events": None if args.parity == "none" else bool(parity_all)},
    }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


def run_e2e_host_only(args, torch, np, fetches, W, world, rank, dist, ctl, wire_bytes, n_pub, gen_s, topic_counts, PartitionedFramedFetches, P):
    """The host side of bytes -> states alone (--host-only): receive + frame every fetch response, device stage stubbed out."""
    eff = effective_cpus()
    threads = max(1, min(args.framing_threads, int(eff[0]) // world - (3 if world == 1 else 1)))
    by_copy = bool(getattr(args, "framing_by_copy", False))
    if dist is not None:
        dist.barrier()
    n_sections = 0
    with PartitionedFramedFetches((f for f, _ in fetches), P, threads=threads, hold=4, overlap=False, device_crc=not by_copy, in_place=not by_copy) as framed:
        t0 = c0 = g0 = None
        for i, (sec, slab) in enumerate(framed):
            n_sections += int(sec.shape[0])
            if i == W - 1 or (W == 0 and i == 0):
                t0, c0, g0 = time.perf_counter(), time.process_time(), framed.cpu_seconds()
        t1, c1, g1 = time.perf_counter(), time.process_time(), framed.cpu_seconds()
        slab_bytes, pinned = framed.slab_bytes()
        counters = framed.counters()
    n_timed = sum(n for _, n in fetches[W:])
    mine = torch.tensor([t1 - t0, c1 - c0, g1[0] - g0[0], g1[1] - g0[1], float(n_timed), float(slab_bytes), float(wire_bytes)], dtype=torch.float64, device=ctl)
    rows = [mine]
    if dist is not None:
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return None
    R = np.array([r.cpu().numpy() for r in rows])
    total_records = float(R[:, 4].sum())
    out = {
        "metric": "records/sec framed by the host (device stage stubbed out)", "value": total_records / float(R[:, 0].max()), "unit": "records/s", "n_gpus": world,
        "steps": len(fetches) - W, "warmup": W, "higher_is_better": True, "data": "synthetic (the e2e topic of this rank's partitions)",
        "config": {"workload": "E2E host side only: receive + frame (in place, CRC-32C left to the device)" if not by_copy else "E2E host side only: framing by copy, CRC-32C on the host",
                   "ranks": world, "cpus_usable": eff[0], "cpu_quota": eff[2], "framing_threads_per_rank": threads, "partitions_per_rank": P // world if world > 1 else P,
                   "per_rank": [{"records": int(r[4]), "wall_s": float(r[0]), "process_cpu_ms_per_1e6_records": float(r[1] * 1e3 / max(1.0, r[4]) * 1e6),
                                 "receive_copy_cpu_ms_per_1e6_records": float(r[2] * 1e3 / max(1.0, r[4]) * 1e6), "framing_cpu_ms_per_1e6_records": float(r[3] * 1e3 / max(1.0, r[4]) * 1e6),
                                 "slab_bytes": int(r[5]), "records_per_s": float(r[4] / r[0])} for r in R],
                   "page_locked_slabs": bool(pinned), "slab_bytes_all_ranks": int(R[:, 5].sum()), "wire_bytes_per_record": float(R[:, 6].sum() / max(1, n_pub * world)) if world == 1 else None,
                   "sections": n_sections, "ingest": counters, "generate_s": gen_s, "topic": topic_counts,
                   "note": "every rank frames its own partitions' fetch responses on its share of the CPUs this process group may use; nothing is pushed to a device, so the figure is the "
                           "host-side ceiling of N ranks on one node — the device side of N > 1 has never been measured (no multi-GPU node was available)"},
    }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


def effective_cpus():
    """CPUs this process can really use: the scheduler affinity, capped by the cgroup CPU quota (the GPU boxes of this
    project show 256 logical CPUs and a quota of 16 — the C restatement scales linearly to 16 threads and not beyond)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2: "<quota|max> <period>"
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.999)))
    return eff, n, quota


def full_log_parity(seg_off, events, gpu_states, threads, slice_events=1 << 28):
    """EVERY aggregate of the HBM-resident log folded by the CPU restatement (oracle/, sequential per aggregate, aggregates
    split over ``threads`` host threads) and compared byte for byte with the GPU's states.  The log is brought to the host
    in slices of whole aggregates (~``slice_events`` events = 4 GiB each), so the host never holds more than one slice."""
    import numpy as np
    import torch

    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth

    t0 = time.perf_counter()
    n = int(seg_off.numel()) - 1
    total = int(seg_off[-1].item())
    cuts = [0, n]
    if total > slice_events:
        targets = torch.arange(slice_events, total, slice_events, device=seg_off.device, dtype=seg_off.dtype)
        cuts += [int(c) for c in torch.searchsorted(seg_off, targets).tolist()]
    cuts = sorted(set(cuts))
    bad, first_bad, checked_events, cpu_s = 0, None, 0, 0.0
    for a0, a1 in zip(cuts[:-1], cuts[1:]):
        e0, e1 = int(seg_off[a0].item()), int(seg_off[a1].item())
        so = (seg_off[a0 : a1 + 1] - e0).cpu().numpy()
        ev = synth.to_event_records(events[e0:e1])
        t1 = time.perf_counter()
        exp = oracle.fold_csr(so, ev, threads=threads)
        cpu_s += time.perf_counter() - t1
        got = gpu_states[a0:a1].cpu().numpy().view(S.STATE_DTYPE).reshape(-1)
        if got.tobytes() != exp.tobytes():
            diff = np.nonzero(got != exp)[0]
            bad += int(diff.size)
            if first_bad is None:
                first_bad = a0 + int(diff[0])
        checked_events += e1 - e0
        del ev, exp, got
    return {"aggregates_checked": n, "events_checked": checked_events, "mismatching_aggregates": bad, "first_mismatch": first_bad,
            "slices": len(cuts) - 1, "seconds": time.perf_counter() - t0, "cpu_fold_seconds": cpu_s, "threads": threads}


def run_cpu_baseline(args, seg_off, events, gpu_states):
    """CPU restatement (oracle/, kind "port") on a bounded sample of the SAME log, all host cores."""
    import torch

    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth

    cores, logical, quota = effective_cpus()
    n_aggs = int(seg_off.numel()) - 1
    # first aggregates of the log holding about 64 M events
    sample_aggs = int(torch.searchsorted(seg_off, torch.tensor([64_000_000], device=seg_off.device))[0])
    sample_aggs = max(1, min(sample_aggs, n_aggs))
    so = seg_off[: sample_aggs + 1].cpu().numpy()
    n_ev = int(so[-1])
    ev = synth.to_event_records(events[:n_ev])
    # parity of the GPU result on the sample, bit for bit
    exp = oracle.fold_csr(so, ev, threads=cores)
    got = gpu_states[:sample_aggs].cpu().numpy().view(S.STATE_DTYPE).reshape(-1)
    parity = got.tobytes() == exp.tobytes()

    def timed(threads, budget_s):
        # threads keep folding their share of the sample `reps` times each (no thread start per pass); reps is sized
        # from a first pass so the leg spends about budget_s of CPU time
        t0 = time.perf_counter()
        oracle.fold_csr_repeated(so, ev, threads, 1)
        one = time.perf_counter() - t0
        reps = int(max(1, min(400, budget_s / max(one * threads, 1e-6))))
        t0 = time.perf_counter()
        oracle.fold_csr_repeated(so, ev, threads, reps)
        return n_ev * reps / (time.perf_counter() - t0)

    all_cores = timed(cores, args.cpu_seconds)
    one_core = timed(1, min(args.cpu_seconds / 4, 4.0))

    # PCIe-inclusive rate (NOT `value`): the same sample handed over as host buffers through
    # surge_replay_load_csr (pageable memory, like a JNI direct buffer), then folded.
    from surge_amd.replay import ReplayEngine

    pcie = None
    try:
        with ReplayEngine(device=gpu_states.device.index or 0) as e2:
            e2.load_csr(so, ev)  # warm-up (allocations)
            t0 = time.perf_counter()
            e2.load_csr(so, ev)
            e2.fold()
            e2.synchronize()
            dt = time.perf_counter() - t0
            pcie = {"events_per_sec": n_ev / dt, "h2d_ms": e2.stats().h2d_ms, "sample_events": n_ev,
                    "GBps": (ev.nbytes + so.nbytes) / dt / 1e9}
    except Exception as exc:  # pragma: no cover
        pcie = {"error": str(exc)}
    full = None
    if args.parity == "full":
        full = full_log_parity(seg_off, events, gpu_states, cores)
    pcie_full = None
    if isinstance(pcie, dict) and pcie.get("GBps"):
        n_all = int(seg_off[-1].item())
        pcie_full = {"seconds": (16 * n_all + 8 * (n_aggs + 1)) / (pcie["GBps"] * 1e9),
                     "note": "the whole log handed over as host buffers at the H2D rate measured on the sample (extrapolated, not run: "
                             "the log is generated on the device)"}
    return {
        "value": all_cores,
        "unit": "events/s",
        "cores": cores,
        "kind": "port",
        "gpu_matches_cpu_full_log": None if full is None else full["mismatching_aggregates"] == 0,
        "full_log_check": full,
        "pcie_inclusive_full_log_estimate": pcie_full,
        "sample": f"first {sample_aggs} aggregates of the same log ({n_ev} events), "
                  f"C restatement of the fold, aggregates split over {cores} host threads "
                  f"({logical} logical CPUs visible, cgroup CPU quota {'none' if quota is None else round(quota, 2)})",
        "host_logical_cpus": logical,
        "cgroup_cpu_quota": quota,
        "single_thread_value": one_core,
        "thread_scaling_efficiency": all_cores / (one_core * cores) if one_core > 0 else None,
        "gpu_matches_cpu_on_sample": parity,
        "pcie_inclusive_gpu": pcie,
    }


if __name__ == "__main__":
    main()
