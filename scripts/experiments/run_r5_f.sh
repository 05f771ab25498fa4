#!/bin/bash
# Round 5, sixth run: the record walks with four-byte loads and the tight chain: ingest GPU tests first, then the kernels one
# push at a time, then the pipeline (hardware queues, framing threads, consumer threads); the C4 shard with 8-event lanes.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5f; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_ingest.log 2>&1; lap "pytest ingest rc=$?"
tail -n 3 $O/pytest_ingest.log
for M in 0 2; do
  SURGE_DBG_DECODE=$M SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05_dbg2_m$M --workload e2e --steps 8 --warmup 2 --txn-flush-events 512 > $O/prof_dbg_m$M.log 2>&1; lap "dbg mode $M rc=$?"
  grep -E "section_kernel|lz4_exec_kernel|lz4_parse_kernel" gpurun_out/prof_r05_dbg2_m$M/*_summary.txt | head -n 3
done
run() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --workload e2e --steps 28 --warmup 2 "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; lap "e2e $name rc=$?"
}
run k512 X=1 -- --txn-flush-events 512
run k512_q8 GPU_MAX_HW_QUEUES=8 -- --txn-flush-events 512
run k512_t12 X=1 -- --txn-flush-events 512 --framing-threads 12
run k512_2t_q8_t12 GPU_MAX_HW_QUEUES=8 -- --txn-flush-events 512 --framing-threads 12 --two-thread-consumer
run k64_q8_t12 GPU_MAX_HW_QUEUES=8 -- --txn-flush-events 64 --framing-threads 12
run k0 X=1 -- --txn-flush-events 0
python - <<'P'
import json
O="gpurun_out/r5f"
for n in ("k512","k512_q8","k512_t12","k512_2t_q8_t12","k64_q8_t12","k0"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "parity", d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"], "fetch_ms", {k: round(v,2) for k,v in c["fetch_ms"].items()}, "framing", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2), "disc %.3g known %.3g"%(c["events_per_s_while_discovering_keys"] or 0, c["events_per_s_all_keys_known"] or 0))
    except Exception as e: print(n, "failed", e)
P
timeout 300 python bench.py --workload c4-shard --no-cpu-baseline > $O/c4shard.json 2> $O/c4shard.err; lap "c4-shard rc=$?"
SURGE_REPLAY_LE_CHUNKED=8 timeout 300 python bench.py --workload c4-shard --no-cpu-baseline > $O/c4shard_le8.json 2> $O/c4shard_le8.err; lap "c4-shard le8 rc=$?"
timeout 300 python bench.py --workload c4-shard --no-cpu-baseline > $O/c4shard_b.json 2> $O/c4shard_b.err; lap "c4-shard rc=$?"
python - <<'P'
import json
O="gpurun_out/r5f"
for n in ("c4shard","c4shard_le8","c4shard_b"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], "frac %.4f"%r["frac"], "kernel_ms", r["kernel_ms_min_median_max"])
    except Exception as e: print(n, "failed", e)
P
