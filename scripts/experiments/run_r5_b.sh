#!/bin/bash
# Round 5, second run: the default line with e2e + c4_shard riding on it; counters (traffic + SQ) for the C4 shard's kernel and
# for the headline SORTED fold; host-side laps of a push.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5b; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; lap "bench rc=$?"
python - <<'P'
import json
O="gpurun_out/r5b"
try:
    d=json.loads([l for l in open(O+"/bench_n1.json") if l.startswith("{")][-1])
    print("default:", d["config"]["algo"], "%.4g"%d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
    print("c4_shard:", json.dumps(d.get("c4_shard"))[:1500])
    e=d.get("e2e",{}); print("e2e:", e.get("value"), e.get("skipped"), json.dumps(e.get("layouts_events_per_s")))
    print("c5", d["c5"].get("value"), "v2", d["v2"].get("roofline",{}).get("frac"))
except Exception as e: print("default failed", e)
P
PROF_PASS_TIMEOUT=120 timeout 700 python scripts/prof_traffic.py r05_c4shard_auto --workload c4-shard > $O/prof_c4shard.log 2>&1; lap "prof c4shard rc=$?"
tail -n 45 $O/prof_c4shard.log
PROF_PASS_TIMEOUT=150 timeout 800 python scripts/prof_traffic.py r05_c3_10Magg_sorted > $O/prof_c3_sorted.log 2>&1; lap "prof c3 sorted rc=$?"
tail -n 45 $O/prof_c3_sorted.log
SURGE_DBG_TIMING=1 timeout 300 python bench.py --workload e2e --steps 8 --warmup 2 > $O/e2e_dbg.json 2> $O/e2e_dbg.err; lap "e2e dbg rc=$?"
grep "surge dbg" $O/e2e_dbg.err | tail -n 40
