#!/bin/bash
# Round 5, fifth run: SORTED pipelined across groups (fold_sorted_pf_kernel) against the plain kernel, A/B/A on one box; where the
# decode stage's kernels spend their time (SURGE_DBG_DECODE modes, one push at a time); the GPU suite.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5e; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none > $O/c3_pf1.json 2> $O/c3_pf1.err; lap "c3 pf rc=$?"
SURGE_REPLAY_SORTED_KERNEL=plain timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none > $O/c3_plain.json 2> $O/c3_plain.err; lap "c3 plain rc=$?"
timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none > $O/c3_pf2.json 2> $O/c3_pf2.err; lap "c3 pf rc=$?"
timeout 300 python bench.py --no-secondary --no-cpu-baseline --parity none --algo chunked > $O/c3_chunked.json 2> $O/c3_chunked.err; lap "c3 chunked rc=$?"
python - <<'P'
import json
O="gpurun_out/r5e"
for n in ("c3_pf1","c3_plain","c3_pf2","c3_chunked"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], r["kernel"], "frac %.4f"%r["frac"], "kernel_ms", r["kernel_ms_min_median_max"], "first fold", d["one_shot"]["first_fold_kernel_ms"], "index", d["one_shot"]["index_build_ms"])
    except Exception as e: print(n, "failed", e)
P
for M in 0 1 2 11 21; do
  SURGE_DBG_DECODE=$M SURGE_BENCH_DEPTH=1 PROF_PASSES=trace PROF_PASS_TIMEOUT=120 timeout 200 python scripts/prof_ingest.py r05_dbg_m$M --workload e2e --steps 8 --warmup 2 --txn-flush-events 512 --parity none > $O/prof_dbg_m$M.log 2>&1; lap "dbg mode $M rc=$?"
  grep -E "section_kernel|lz4_exec_kernel|lz4_parse_kernel" gpurun_out/prof_r05_dbg_m$M/*_summary.txt | head -n 3
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; lap "pytest rc=$?"
tail -n 4 $O/pytest.log
