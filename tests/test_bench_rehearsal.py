"""bench.py's N > 1 path, run for real on a one-GPU box: two ranks on cuda:0 (SURGE_BENCH_REHEARSAL=1: gloo control plane),
the snapshot exchange through the C ABI over tests/rccl_stub (RCCL itself refuses two ranks on one device).  A functional
check of code the 8-GPU driver run depends on — shard generation by Kafka partition, the overlapped exchange, the gathered
snapshot's verification, the JSON contract — not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

from test_comm import build_rccl_stub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(tmp_path, extra, world=2, launcher="self"):
    env = dict(os.environ, SURGE_BENCH_REHEARSAL="1", SURGE_RCCL_LIBRARY=build_rccl_stub(), SURGE_RCCL_STUB_DIR=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if launcher == "self":  # the way the driver invokes it: a plain command, bench.py starts its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra,algo", [(["--aggregates", "400000"], None), (["--workload", "c2", "--aggregates", "200000"], "rows")])
def test_two_rank_bench_rehearsal_on_one_gpu(tmp_path, extra, algo):
    d = run_bench(tmp_path, extra)
    cfg = d["config"]
    if algo is None:  # the Zipf log defaults to the fold AUTO picks straight from the CSR log: no copy of the log, only its index
        assert cfg["algo"] in ("chunked", "sorted", "flat") and d["one_shot"]["tile_major_copy_bytes"] == 0 and d["one_shot"]["relayout_ms"] == 0
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "strong" and "rehearsal" in d
    assert d["metric"] == "events/sec replayed" and d["unit"] == "events/s" and d["higher_is_better"] is True
    assert len(cfg["per_rank_events"]) == 2 and sum(cfg["per_rank_events"]) == cfg["events"] and min(cfg["per_rank_events"]) > 0
    assert "C ABI" in cfg["exchange"] and "FALLBACK" not in cfg["exchange"] and cfg["exchange_alone_ms"] > 0
    assert abs(d["value"] - cfg["events"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0 and d["cpu_baseline"] is None
    if algo:
        assert algo in cfg["algo"].lower()


@pytest.mark.gpu
def test_bench_rehearsal_falls_back_to_the_torch_exchange_and_says_so(tmp_path):
    d = run_bench(tmp_path, ["--aggregates", "200000", "--gather", "torch"], launcher="torchrun")  # the explicit launcher still works
    assert "torch.distributed" in d["config"]["exchange"] and d["config"]["exchange_alone_ms"] > 0


@pytest.mark.gpu
def test_eight_rank_bench_rehearsal_has_the_shape_of_config_c4(tmp_path):
    """World size 8 — the driver's scaling run — with a small log: 64 Kafka partitions over 8 ranks (8 partitions each),
    every rank exchanges with 7 peers per step."""
    d = run_bench(tmp_path, ["--aggregates", "320000"], world=8)
    cfg = d["config"]
    assert d["n_gpus"] == 8 and len(cfg["per_rank_events"]) == 8 and sum(cfg["per_rank_events"]) == cfg["events"]
    assert len(cfg["per_rank_fold_kernel_ms"]) == 8 and min(cfg["per_rank_fold_kernel_ms"]) > 0
    assert "C4:" in cfg["workload"] and "% 8" in cfg["workload"] and "C ABI" in cfg["exchange"]
    # shards by partitionForKey(id, 64) % 8 are balanced to a few percent at this size
    assert max(cfg["per_rank_events"]) < 1.15 * min(cfg["per_rank_events"])


def run_single(args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_bench_line_carries_the_tile_major_fold_c2_c5_and_v2_beside_the_headline():
    """The default invocation at a small size: the headline is the fold AUTO picks from the CSR log, `tile_major` the fold over
    the re-laid copy with its one-off cost, `secondary` config C2 (AUTO + tile-major), `c5` / `v2` the streaming config and
    the ABI v2 path — all timed by whoever runs bench.py, not only by a builder's own invocations."""
    d = run_single(["--aggregates", "200000", "--steps", "5", "--warmup", "2", "--cpu-seconds", "2"])
    assert d["config"]["algo"] in ("chunked", "sorted", "flat") and d["csr_direct"] is None
    tm = d["tile_major"]
    assert tm["algo"] == "tiled" and tm["states_equal_primary"] is True and tm["one_shot"]["relayout_ms"] > 0 and tm["frac"] > 0
    sec = d["secondary"]
    assert sec["config"]["algo"] == "rows" and sec["tile_major"]["algo"] == "tiled" and sec["tile_major"]["states_equal_primary"] is True
    assert d["c5"]["cpu_baseline"]["gpu_matches_cpu_full_run"] is True and d["c5"]["value"] > 0
    assert d["v2"]["cpu_baseline"]["gpu_matches_cpu_full_log"] is True and d["v2"]["roofline"]["kernel"] == "surge_slots_tiled2"
    assert d["roofline"]["stream_read_probe_GBps"] > 1000 and d["cpu_baseline"]["gpu_matches_cpu_full_log"] is True


@pytest.mark.gpu
def test_bench_workload_v2_reports_four_variants_of_the_same_device_code_and_full_log_parity():
    d = run_single(["--workload", "v2", "--aggregates", "30000", "--steps", "3", "--warmup", "1"])
    v = d["config"]["variants"]
    assert set(v) == {"specialised/tiled", "specialised/csr", "interpreter/tiled", "interpreter/csr"}
    assert all(x["states_equal_first_variant"] for x in v.values()) and d["cpu_baseline"]["gpu_matches_cpu_full_log"] is True
    assert d["roofline"]["kernel"] == "surge_slots_tiled2" and d["one_shot"]["schema_compile_ms"] >= 0 and "hiprtc" in d["config"]["kernels"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--serial-framing"], ["--codec", "none"], ["--no-capacity-hint"], ["--two-thread-consumer"]])
def test_bench_workload_e2e_goes_from_topic_bytes_to_states_and_checks_them_against_the_source_events(extra):
    """The C3-shaped topic at a small size: 30 000 aggregates over 64 partitions, lz4 batches of 16 KiB, fetches of 20 000
    records framed per partition on host threads, one device push per fetch (four in flight; ``--two-thread-consumer``:
    enqueued by a worker thread while this thread interns and folds the oldest without a host wait),
    states compared with the oracle's fold of the events the GENERATOR published — not of what the device decoded."""
    d = run_single(["--workload", "e2e", "--aggregates", "30000", "--batch-events", "20000", "--warmup", "1", "--framing-threads", "4"] + extra)
    cfg = d["config"]
    assert cfg["fetch_records"] == 20000 and cfg["partitions"] == 64 and cfg["keys_interned"] == 30000
    assert cfg["decoder"]["records_delivered"] == cfg["ingest"]["records_delivered"] == cfg["events_timed"] + 20000  # + the warm-up fetch
    assert cfg["pushes_in_flight"] == (1 if "--serial-framing" in extra else 4) and cfg["decoder"]["hash_reseeds"] == 0
    assert cfg["consumer"].startswith("push worker thread" if "--two-thread-consumer" in extra else "one thread")
    assert d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"] is True and d["value"] > 0 and d["n_gpus"] == 1
    assert abs(d["value"] - cfg["events_timed"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--bound-log"], ["--framing-by-copy"]])
def test_bench_workload_e2e_mixed_topic_bank_account_events_with_headers_and_long_rows(extra):
    """VERDICT r5 item 7, the wider topic: BankAccount events (BankAccountSurgeModel.scala:26-32 — UUID record keys without
    ':', no sequence numbers, Double balances as play-json text), two headers on every record, every 128th account publishing
    64..256 events (a late fetch holds the same accounts dozens of times) — states equal to the oracle's fold of the source
    events, through the pipelined fold-per-fetch path, the fold-once packer and round 5's by-copy framing."""
    d = run_single(["--workload", "e2e", "--e2e-topic", "mixed", "--aggregates", "60000", "--batch-events", "20000", "--warmup", "1", "--framing-threads", "3"] + extra)
    cfg = d["config"]
    assert cfg["topic_model"] == "mixed" and cfg["record_header_bytes"] > 80 and 200 <= cfg["max_events_of_one_aggregate"] <= 256
    assert cfg["keys_interned"] == 60000 and cfg["decoder"]["records_delivered"] == cfg["ingest"]["records_delivered"] == cfg["events_timed"] + 20000
    assert cfg["wire_bytes_per_record"] > 40 and cfg["decoder"]["doubles_parsed_on_host"] == 0
    assert d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"] is True and d["value"] > 0
    if "--bound-log" in extra:
        assert cfg["bound_log"]["staged_events"] == cfg["decoder"]["records_delivered"] and cfg["bound_log"]["refold_events_per_s"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_bench_workload_e2e_shards_the_ingest_by_partition_over_the_ranks(tmp_path, world):
    """`--workload e2e --gpus N` — every rank frames and decodes the partitions p % N == rank, folds its
    shard, and the final snapshot is all-gathered through the C ABI (here over the stub transport, all ranks on one GPU):
    one line with n_gpus = N, per-rank records, every rank's shard equal to the oracle's fold of its source events, every
    block of the gathered snapshot equal to its owner's shard."""
    d = run_bench(tmp_path, ["--workload", "e2e", "--aggregates", "40000", "--batch-events", "8000", "--framing-threads", "2"] + (["--e2e-topic", "mixed"] if world == 2 else []), world=world)
    cfg = d["config"]
    assert d["n_gpus"] == world and len(cfg["per_rank_events"]) == world and min(cfg["per_rank_events"]) > 0
    assert sum(cfg["per_rank_events"]) == cfg["events_timed"] and cfg["keys_interned"] == 40000 == cfg["gathered_aggregates"]
    assert cfg["snapshot_exchange_ms"] > 0 and f"p % {world} == rank" in cfg["parallelism"]
    assert d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"] is True and "REHEARSAL" in cfg["workload"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--host-framing"], ["--device-batches"]])
def test_bench_workload_c5_streams_micro_batches_publishes_deltas_and_checks_the_whole_run(extra):
    """BASELINE config 5 as the driver can start it, at a small size: micro-batches onto the resident state, a state-topic
    delta every 10 batches (framed on the device, or by the host writer), the resident state after the run equal to the
    oracle's replay of the same batches."""
    d = run_single(["--workload", "c5", "--aggregates", "50000", "--batch-events", "5000", "--steps", "40", "--warmup", "3", "--snapshot-every", "10"] + extra)
    cfg = d["config"]
    assert cfg["snapshot_ms"]["n"] == 4 and cfg["snapshot_published_aggregates_mean"] > 1000 and cfg["snapshot_record_batch_bytes_mean"] > 50_000
    assert cfg["snapshot_framing"].startswith("host" if "--host-framing" in extra else "device")
    assert ("device_framing_copy_crc_ms" in cfg["snapshot_parts_ms_mean"]) == ("--host-framing" not in extra)
    assert d["cpu_baseline"]["gpu_matches_cpu_full_run"] is True and d["value"] > 0 and d["roofline"]["kernel"].startswith("fold_kernel<FLAT")
