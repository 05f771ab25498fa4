#!/usr/bin/env python3
"""bench.py — events/sec replayed by the GPU aggregate fold (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload at every N: BASELINE.json ``configs[1]`` (SURVEY §8d "C2") PER GPU — 1,000,000
aggregates x 256 fixed-width events (16 B events, 64 B state), i.e. weak scaling: the global log has
N x 1M aggregates, sharded by the reference's Kafka partitioner (murmur3(id) % 64 -> gpu =
partition % N, KafkaPartitioner.scala:8).  One "step" = one full replay of the rank's HBM-resident
shard (plan + fold kernel) and, for N > 1, the RCCL all-gather of the final snapshot, overlapped
with the next step's fold on a side stream.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  ``roofline`` prices the fold kernel against the 8 TB/s HBM peak using
the algorithmic bytes 16*E + 8*(A+1) + 64*A and the kernel's HIP-event time measured
inside the timed region on the launch stream.  ``cpu_baseline`` (rank 0, N == 1 only) times the CPU
restatement (oracle/, "port") on a bounded sample of the same log on this box's host cores, and
the GPU result for that sample is checked bit-for-bit against it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL between processes needs this (already exported on the GPU boxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
AGG_PER_GPU = 1_000_000
EVENTS_PER_AGG = 256
N_PARTITIONS = 64
SEED = 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--aggregates", type=int, default=AGG_PER_GPU, help="aggregates per GPU (default = config C2)")
    ap.add_argument("--events-per-aggregate", type=int, default=EVENTS_PER_AGG)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 fixed, 2 flat, 3 rows, 4 sorted")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"],
                    help="c2 (default, the config the metric is quoted on): fixed fan-in; c3: Zipf(1..4096) event counts")
    ap.add_argument("--zipf-aggregates", type=int, default=10_000_000, help="aggregates per GPU for --workload c3")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from surge_amd import schema as S
    from surge_amd import synth
    from surge_amd.replay import ReplayEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the replay engine has no CPU fallback")
    # rehearsal of the N > 1 control flow on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL
    rehearsal = os.environ.get("SURGE_BENCH_SINGLE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    A, L = args.aggregates, args.events_per_aggregate
    zipf = args.workload == "c3"
    if zipf:
        A = args.zipf_aggregates
    eng = ReplayEngine(device=local_rank)
    compute = torch.cuda.Stream(device=dev)
    eng.use_stream(compute)

    # ---- build this rank's HBM-resident shard -----------------------------------------------------
    if world == 1:
        agg_ids = torch.arange(A, dtype=torch.int64, device=dev) if zipf else None
        n_local = A
    else:
        from surge_amd.dist import SnapshotGather, local_aggregate_ids

        agg_ids = local_aggregate_ids(A * world, N_PARTITIONS, rank, world, dev, eng)
        n_local = int(agg_ids.numel())
    if zipf:
        # the rank's aggregates keep the event counts and contents they have in the global log
        lens = synth.zipf_lengths(agg_ids, 3)
        seg_off, events = synth.csr_log_device(lens, 3, agg_ids=agg_ids,
                                               global_seg_off=_LazyGlobalOffsets(agg_ids, lens))
        n_events_local = int(seg_off[-1].item())
    elif world == 1:
        seg_off, events = synth.fixed_log_device(A, L, SEED, dev)
        n_events_local = n_local * L
    else:
        seg_off, events = synth.fixed_log_for_aggregates_device(agg_ids, L, SEED)
        n_events_local = n_local * L
    torch.cuda.synchronize(dev)

    if world == 1:
        bufs = [torch.zeros((n_local, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
        gather = None
    else:
        gather = SnapshotGather(n_local, dev, engine=eng)
        bufs = gather.make_local_buffers()
    eng.load_csr(seg_off, events, None, bufs[0])

    fold_done = [torch.cuda.Event(), torch.cuda.Event()]

    def step(i):
        slot = i & 1
        if gather is not None:
            gather.wait(slot, compute)  # the gather that last read bufs[slot] must be finished
        eng.set_state_out(bufs[slot])
        eng.fold(args.algo)
        if gather is not None:
            fold_done[slot].record(compute)
            gather.launch(slot, bufs[slot], fold_done[slot])

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_(t, op):
        if dist is None:
            return t
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            return h.to(dev)
        dist.all_reduce(t, op=op)
        return t

    for i in range(args.warmup):
        step(i)
    sync_all()
    eng.stats_reset()

    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    totals = torch.tensor([n_events_local, n_local], dtype=torch.int64, device=dev)
    if dist is not None:
        elapsed = reduce_(elapsed, dist.ReduceOp.MAX)
        totals = reduce_(totals, dist.ReduceOp.SUM)
    elapsed_s = float(elapsed.item())
    total_events, total_aggs = int(totals[0].item()), int(totals[1].item())

    st = eng.stats()
    kernel_ms = st.sum_fold_kernel_ms / max(st.timed_folds, 1)
    traffic, traffic_src = pmc_traffic(args, st.last_algo)
    achieved = st.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # ---- extras on rank 0 ---------------------------------------------------------------------------
    result = None
    if rank == 0:
        probe_gbps = None
        try:
            ms = min(eng.stream_probe_ms(events) for _ in range(5))
            probe_gbps = events.numel() * 8 / (ms * 1e-3) / 1e9
        except Exception as e:  # pragma: no cover
            print(f"stream probe failed: {e}", file=sys.stderr)
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = run_cpu_baseline(args, seg_off, events, bufs[(args.warmup + args.steps - 1) & 1], L)
        ms_per_step = elapsed_s / args.steps * 1e3
        if world > 1:
            # the gathered snapshot must contain this rank's own shard, bit for bit
            last = (args.warmup + args.steps - 1) & 1
            torch.cuda.synchronize(dev)
            own = gather.result(last)[rank, :n_local]
            assert torch.equal(own, bufs[last][:n_local]), "all-gathered snapshot does not contain the local shard"
        result = {
            "metric": "events/sec replayed",
            "value": total_events * args.steps / elapsed_s,
            "unit": "events/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32/int64 adds + bit-copied f64",
            "data": "synthetic (counter-hash log, seed 2; surge_amd/synth.py)",
            "aggregates_per_sec": total_aggs * args.steps / elapsed_s,
            "config": {
                "workload": (f"C3 per GPU: {A} aggregates, Zipf(1..4096) events each ({n_events_local} events on rank 0), CSR, "
                             "16 B events, 64 B state, log resident in HBM") if zipf else
                f"C2 per GPU: {A} aggregates x {L} events, 16 B events, 64 B state, log resident in HBM",
                "aggregates_per_gpu": A,
                "events_per_aggregate": "zipf(1..4096), mean ~460" if zipf else L,
                "algo": {S.ALGO_FIXED: "fixed", S.ALGO_FLAT: "flat", S.ALGO_ROWS: "rows", S.ALGO_SORTED: "sorted"}.get(st.last_algo, str(st.last_algo)),
                "wave_tasks": st.n_tasks,
                "sharding": "single shard" if world == 1 else
                f"murmur3(acct-%08d) % {N_PARTITIONS} -> gpu = partition % {world}; final snapshot exchanged over "
                f"{'gloo (single-GPU rehearsal)' if rehearsal else 'RCCL'} ({'grouped per-peer send/recv' if gather.mode == 'p2p' else 'all_gather_into_tensor'}, "
                f"{'40-byte wire form' if gather.packed else '64-byte states'}) on a side stream, overlapped with the next fold",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": {S.ALGO_FIXED: "fold_kernel<FIXED,16>", S.ALGO_FLAT: "fold_kernel<FLAT,16>",
                           S.ALGO_ROWS: "fold_rows_kernel<8>", S.ALGO_SORTED: "fold_sorted_kernel<16>"}.get(st.last_algo, "?"),
                "kernel_ms": kernel_ms,
                "algorithmic_bytes": st.algorithmic_bytes,
                "timed_launches": st.timed_folds,
                "stream_read_probe_GBps": probe_gbps,
            },
            "cpu_baseline": cpu_baseline,
        }
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def pmc_traffic(args, algo):
    """HBM bytes per launch from the rocprofv3 PMC passes of THIS command (separate --pmc runs for
    FETCH_SIZE and WRITE_SIZE, gfx950 correction x2 on FETCH_SIZE, KB -> bytes), as committed under
    profiles/ by scripts/prof.sh.  bench.py cannot collect PMC counters itself; null when there is no
    profile for the workload/kernel being run."""
    if args.workload == "c2" and args.aggregates == AGG_PER_GPU and args.events_per_aggregate == EVENTS_PER_AGG:
        name, want = "r01_final_c2_rows_summary.txt", {3: "fold_rows"}.get(algo)
    elif args.workload == "c3" and args.zipf_aggregates == 10_000_000:
        name, want = "r01_final_c3_sorted16_10Magg_summary.txt", {4: "fold_sorted"}.get(algo)
    else:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    if want is None or not os.path.exists(path):
        return None, None
    fetch = write = None
    for line in open(path):
        parts = line.split()
        if len(parts) >= 5 and parts[1] == want and parts[2] == "FETCH_SIZE":
            fetch = float(parts[-1].split("=")[1])
        if len(parts) >= 5 and parts[1] == want and parts[2] == "WRITE_SIZE":
            write = float(parts[-1].split("=")[1])
    if fetch is None or write is None:
        return None, None
    return fetch * 1024 * 2 + write * 1024, f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)"


class _LazyGlobalOffsets:
    """``global_seg_off[agg]`` for the sharded Zipf log without materialising the global prefix sum:
    only the event *hash index* needs to be unique per (aggregate, position), so aggregate a's events
    are numbered from a * 4096 (4096 = the longest possible segment)."""

    def __init__(self, agg_ids, lens):
        pass

    def __getitem__(self, agg):
        return agg * 4096


def run_cpu_baseline(args, seg_off, events, gpu_states, L):
    """CPU restatement (oracle/, kind "port") on a bounded sample of the SAME log, all host cores."""
    import numpy as np

    from oracle import oracle
    from surge_amd import schema as S
    from surge_amd import synth

    import torch

    cores = os.cpu_count() or 1
    n_aggs = int(seg_off.numel()) - 1
    # first aggregates of the log holding about 64 M events (250k aggregates of config C2)
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
        oracle.fold_csr(so, ev, threads=threads)  # warm-up pass
        reps, t0 = 0, time.perf_counter()
        while True:
            oracle.fold_csr(so, ev, threads=threads)
            reps += 1
            dt = time.perf_counter() - t0
            if dt * threads >= budget_s or reps >= 200:
                return n_ev * reps / dt

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
    return {
        "value": all_cores,
        "unit": "events/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {sample_aggs} aggregates of the same log ({n_ev} events), "
                  f"C restatement of the fold, aggregates split over {cores} host threads",
        "single_thread_value": one_core,
        "gpu_matches_cpu_on_sample": parity,
        "pcie_inclusive_gpu": pcie,
    }


if __name__ == "__main__":
    main()
