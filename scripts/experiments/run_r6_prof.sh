#!/bin/bash
# Round 6: kernel trace + FETCH / WRITE / SQ counter passes of the headline fold, the C4 shard's, C2's and the tile-major folds on the final sources
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6prof; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
PROF_PASS_TIMEOUT=150 timeout 800 python scripts/prof_traffic.py r06_c3_10Magg_sorted --parity none > $O/prof_c3_sorted.log 2>&1; lap "prof c3 sorted rc=$?"
grep -E "fold_sorted_pf|traffic per launch" $O/prof_c3_sorted.log | cut -c1-200
PROF_PASS_TIMEOUT=100 timeout 500 python scripts/prof_traffic.py r06_c4shard_auto --workload c4-shard > $O/prof_c4shard.log 2>&1; lap "prof c4shard rc=$?"
grep -E "fold_chunked|traffic per launch" $O/prof_c4shard.log | cut -c1-200
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=100 timeout 400 python scripts/prof_traffic.py r06_c2_rows --workload c2 > $O/prof_c2_rows.log 2>&1; lap "prof c2 rows rc=$?"
grep -E "fold_rows|traffic per launch" $O/prof_c2_rows.log | cut -c1-200
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=150 timeout 600 python scripts/prof_traffic.py r06_c3_10Magg_tiled --algo tiled --parity none > $O/prof_c3_tiled.log 2>&1; lap "prof c3 tiled rc=$?"
grep -E "fold_tiled|traffic per launch" $O/prof_c3_tiled.log | cut -c1-200
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=100 timeout 400 python scripts/prof_traffic.py r06_c2_tiled --workload c2 --algo tiled > $O/prof_c2_tiled.log 2>&1; lap "prof c2 tiled rc=$?"
grep -E "fold_tiled|traffic per launch" $O/prof_c2_tiled.log | cut -c1-200
du -sh gpurun_out
