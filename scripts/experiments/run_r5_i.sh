#!/bin/bash
# Round 5, ninth run: the GPU suite on the round's sources (key table in 16-byte slots, three stage-1 streams, onesweep group-by,
# LZ4 launches re-classed); the bytes -> states path, same box: three streams against a stream per slot (with 4 and 8 hardware
# queues), onesweep against rocPRIM's merge sort, the LZ4 LDS layouts; one push at a time under the kernel trace; the counters
# (traffic + SQ) of the headline fold (fold_sorted_pf_kernel) and of the C4 shard's (fold_chunked_kernel).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5i; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; lap "pytest rc=$?"
tail -n 5 $O/pytest.log
run() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --workload e2e --warmup 2 --parity none "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; lap "e2e $name rc=$?"
}
run k512_a SURGE_BENCH_TRACE=1 -- --steps 28 --txn-flush-events 512
run k512_s5 SURGE_INGEST_PUSH_STREAMS=5 SURGE_BENCH_TRACE=1 -- --steps 28 --txn-flush-events 512
run k512_s5q8 SURGE_INGEST_PUSH_STREAMS=5 GPU_MAX_HW_QUEUES=8 -- --steps 28 --txn-flush-events 512
run k512_merge SURGE_REPLAY_GROUPBY_SORT=merge -- --steps 28 --txn-flush-events 512
run k512_lz4old SURGE_INGEST_LZ4_PAD=64 SURGE_INGEST_LZ4_PARSE_CLASSES=2 -- --steps 28 --txn-flush-events 512
run k512_b X=1 -- --steps 28 --txn-flush-events 512
run k512_q8 GPU_MAX_HW_QUEUES=8 -- --steps 28 --txn-flush-events 512
run k64 X=1 -- --steps 10 --txn-flush-events 64
run k0 X=1 -- --steps 10 --txn-flush-events 0
python - <<'P'
import json
O="gpurun_out/r5i"
for n in ("k512_a","k512_s5","k512_s5q8","k512_merge","k512_lz4old","k512_b","k512_q8","k64","k0"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "fetch_ms", {k: round(v,2) for k,v in c["fetch_ms"].items()}, "framing", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2))
    except Exception as e: print(n, "failed", e)
P
grep "ms between completed folds" $O/e2e_k512_a.err $O/e2e_k512_s5.err
grep "\[bench\] fetch" $O/e2e_k512_a.err | awk '{print $3, $5, $9}' | tr '\n' ';'; echo
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace,sq1,sq2 PROF_PASS_TIMEOUT=120 timeout 400 python scripts/prof_ingest.py r05_e2e_k512_depth1 --workload e2e --steps 10 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_d1.log 2>&1; lap "prof depth1 rc=$?"
head -n 24 gpurun_out/prof_r05_e2e_k512_depth1/*_summary.txt
PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05_e2e_k512 --workload e2e --steps 12 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_d4.log 2>&1; lap "prof depth4 rc=$?"
head -n 16 gpurun_out/prof_r05_e2e_k512/*_summary.txt
PROF_PASS_TIMEOUT=150 timeout 800 python scripts/prof_traffic.py r05_c3_10Magg_sorted --parity none > $O/prof_c3_sorted.log 2>&1; lap "prof c3 sorted rc=$?"
tail -n 42 $O/prof_c3_sorted.log
PROF_PASS_TIMEOUT=120 timeout 700 python scripts/prof_traffic.py r05_c4shard_auto --workload c4-shard > $O/prof_c4shard.log 2>&1; lap "prof c4shard rc=$?"
tail -n 42 $O/prof_c4shard.log
