set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6prof; mkdir -p $O
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=200 timeout 800 python scripts/prof_traffic.py r06_c3_10Magg_sorted_fw --parity none > $O/prof_c3_sorted_fw.log 2>&1; echo "rc=$?"
grep -E "fold_sorted_pf|traffic per launch|killed" $O/prof_c3_sorted_fw.log | cut -c1-200
