#!/bin/bash
# Round 5, eighth run: the record kernel with workgroups as wide as the push's largest batch (64 / 128 / 192 / 256 lanes, 80 VGPRs,
# three LDS classes) against round 4's shape (256 lanes, two classes), one push at a time and pipelined, on the three topic layouts;
# where the slow fetches of the K = 512 layout come from (per-fetch trace); the folds: C3 through the chunk table with nothing cut
# (= SORTED with its row table gathered at index time) against the pipelined SORTED kernel; the C4 shard over chunk length, lane
# events and kernel.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5h; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -m gpu -x -q > $O/pytest_ingest.log 2>&1; lap "pytest ingest rc=$?"
tail -n 3 $O/pytest_ingest.log
# one push at a time: what each kernel costs alone
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05h_k512_d1_new --workload e2e --steps 10 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_new.log 2>&1; lap "prof new rc=$?"
SURGE_INGEST_SEC_THREADS=256 SURGE_INGEST_SEC_CLASSES=2 SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05h_k512_d1_old --workload e2e --steps 10 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_old.log 2>&1; lap "prof old rc=$?"
SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05h_k64_d1_new --workload e2e --steps 10 --warmup 2 --txn-flush-events 64 --parity none > $O/prof_k64_new.log 2>&1; lap "prof k64 new rc=$?"
SURGE_INGEST_SEC_THREADS=256 SURGE_INGEST_SEC_CLASSES=2 SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05h_k64_d1_old --workload e2e --steps 10 --warmup 2 --txn-flush-events 64 --parity none > $O/prof_k64_old.log 2>&1; lap "prof k64 old rc=$?"
for t in r05h_k512_d1_new r05h_k512_d1_old r05h_k64_d1_new r05h_k64_d1_old; do echo "== $t"; grep -E "section_kernel|lz4_exec_kernel|lz4_parse_kernel|copyBuffer " gpurun_out/prof_$t/*_summary.txt | head -n 4; done
run() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --workload e2e --warmup 2 "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; lap "e2e $name rc=$?"
}
run k512_new SURGE_BENCH_TRACE=1 -- --steps 28 --txn-flush-events 512 --parity none
run k512_old SURGE_INGEST_SEC_THREADS=256 SURGE_INGEST_SEC_CLASSES=2 -- --steps 28 --txn-flush-events 512 --parity none
run k512_new2 X=1 -- --steps 28 --txn-flush-events 512 --parity none
run k64_new X=1 -- --steps 10 --txn-flush-events 64 --parity none
run k64_old SURGE_INGEST_SEC_THREADS=256 SURGE_INGEST_SEC_CLASSES=2 -- --steps 10 --txn-flush-events 64 --parity none
run k0_new X=1 -- --steps 10 --txn-flush-events 0 --parity none
python - <<'P'
import json
O="gpurun_out/r5h"
for n in ("k512_new","k512_old","k512_new2","k64_new","k64_old","k0_new"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "fetch_ms", {k: round(v,2) for k,v in c["fetch_ms"].items()}, "framing", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2))
    except Exception as e: print(n, "failed", e)
P
grep "^\[bench\] [mfprw]" $O/e2e_k512_new.err
grep "\[bench\] fetch" $O/e2e_k512_new.err | awk '{print $3, $5, $7, $9}' | tr '\n' ';'; echo
fold() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none "$@" > $O/$name.json 2> $O/$name.err; lap "$name rc=$?"
}
fold c3_pf X=1 --
fold c3_chunked X=1 -- --algo chunked
fold c3_pf2 X=1 --
fold sh_default X=1 -- --workload c4-shard
fold sh_t1536 SURGE_REPLAY_CHUNK_T=1536 -- --workload c4-shard
fold sh_t3072 SURGE_REPLAY_CHUNK_T=3072 -- --workload c4-shard
fold sh_t4104 SURGE_REPLAY_CHUNK_T=4104 -- --workload c4-shard --algo chunked
fold sh_le8 SURGE_REPLAY_LE_CHUNKED=8 -- --workload c4-shard
fold sh_sorted X=1 -- --workload c4-shard --algo sorted
fold sh_default2 X=1 -- --workload c4-shard
python - <<'P'
import json
O="gpurun_out/r5h"
for n in ("c3_pf","c3_chunked","c3_pf2","sh_default","sh_t1536","sh_t3072","sh_t4104","sh_le8","sh_sorted","sh_default2"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], r["kernel"], "frac %.4f"%r["frac"], "kernel_ms", [round(x,4) for x in r["kernel_ms_min_median_max"]], "index", (d.get("one_shot") or {}).get("index_build_ms"))
    except Exception as e: print(n, "failed", e)
P
