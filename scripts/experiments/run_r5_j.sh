#!/bin/bash
# Round 5, tenth run (short): the small-flush layout is bound by the host framing (2.1 - 2.6 ms per fetch on 8 threads): 8 against
# 12 framing threads on one box; the 0.1 M-aggregate Zipf log (0.74 GB) through FLAT (AUTO's pick), CHUNKED and SORTED.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5j; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
run() { # name, args
  local name=$1; shift
  timeout 300 python bench.py --workload e2e --warmup 2 --parity none "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; lap "e2e $name rc=$?"
}
run k64_t8 --steps 10 --txn-flush-events 64
run k64_t12 --steps 10 --txn-flush-events 64 --framing-threads 12
run k64_t8b --steps 10 --txn-flush-events 64
run k64_t14 --steps 10 --txn-flush-events 64 --framing-threads 14
run k512_t12 --steps 28 --txn-flush-events 512 --framing-threads 12
run k512_t8 --steps 28 --txn-flush-events 512
python - <<'P'
import json
O="gpurun_out/r5j"
for n in ("k64_t8","k64_t12","k64_t8b","k64_t14","k512_t12","k512_t8"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "fetch_ms", {k: round(v,2) for k,v in c["fetch_ms"].items()}, "framing", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2))
    except Exception as e: print(n, "failed", e)
P
fold() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-secondary --no-cpu-baseline --parity none --workload c3 --aggregates 100000 --steps 50 "$@" > $O/$name.json 2> $O/$name.err; lap "$name rc=$?"
}
fold z100k_auto X=1 --
fold z100k_chunked X=1 -- --algo chunked
fold z100k_chunked_t512 SURGE_REPLAY_CHUNK_T=512 -- --algo chunked
fold z100k_chunked_t1024 SURGE_REPLAY_CHUNK_T=1024 -- --algo chunked
fold z100k_sorted X=1 -- --algo sorted
fold z100k_auto2 X=1 --
python - <<'P'
import json
O="gpurun_out/r5j"
for n in ("z100k_auto","z100k_chunked","z100k_chunked_t512","z100k_chunked_t1024","z100k_sorted","z100k_auto2"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], r["kernel"], "frac %.4f"%r["frac"], "kernel_ms", [round(x,4) for x in r["kernel_ms_min_median_max"]])
    except Exception as e: print(n, "failed", e, open(f"{O}/{n}.err").read()[-300:])
P
