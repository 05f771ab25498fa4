#!/bin/bash
# Round 5, eleventh run: whole-aggregate groups of SORTED / CHUNKED walk concrete states (apply_event_concrete, 7 VALU fewer per
# event): parity (the kernel fuzz, the full-size C3 / C4-shard logs against the oracle), the concrete against the transformer walk
# on one box (A/B/A), the traffic counters of the new sources.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5k; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=8 > $O/pytest_folds.log 2>&1; lap "pytest folds rc=$?"
tail -n 14 $O/pytest_folds.log
fold() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none "$@" > $O/$name.json 2> $O/$name.err; lap "$name rc=$?"
}
fold c3_conc X=1 --
fold c3_trans SURGE_REPLAY_WALK=transformer --
fold c3_conc2 X=1 --
fold c3_trans2 SURGE_REPLAY_WALK=transformer --
fold sh_conc X=1 -- --workload c4-shard
fold sh_trans SURGE_REPLAY_WALK=transformer -- --workload c4-shard
fold sh_conc2 X=1 -- --workload c4-shard
fold sh_trans2 SURGE_REPLAY_WALK=transformer -- --workload c4-shard
python - <<'P'
import json
O="gpurun_out/r5k"
for n in ("c3_conc","c3_trans","c3_conc2","c3_trans2","sh_conc","sh_trans","sh_conc2","sh_trans2"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], r["kernel"], "frac %.4f"%r["frac"], "kernel_ms", [round(x,4) for x in r["kernel_ms_min_median_max"]])
    except Exception as e: print(n, "failed", e)
P
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=150 timeout 500 python scripts/prof_traffic.py r05_c3_10Magg_sorted --parity none > $O/prof_c3_sorted.log 2>&1; lap "prof c3 sorted rc=$?"
tail -n 4 $O/prof_c3_sorted.log
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=120 timeout 400 python scripts/prof_traffic.py r05_c4shard_auto --workload c4-shard > $O/prof_c4shard.log 2>&1; lap "prof c4shard rc=$?"
tail -n 4 $O/prof_c4shard.log
