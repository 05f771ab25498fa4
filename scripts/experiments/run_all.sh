# full verification + the profiles the traffic manifest is built from (one gpurun call); every step bounded.
# PROF_SET: which configurations to re-profile (default: all five)
mkdir -p gpurun_out/final
rm -rf gpurun_out/prof_r02_*
timeout 900 python -m pytest tests -q -m gpu -rs > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
grep -E "passed|failed|error|rc=" gpurun_out/final/pytest.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; echo "bench rc=$?"
ZIPF_AGGS=1250000 FOLDS=40 timeout 200 python scripts/experiments/variance_probe.py 2>&1 | grep "^rep"
export PROF_PASS_TIMEOUT=200
set -- ${PROF_SET:-c4s c2 c3 300k 100k}
for cfg in "$@"; do
case $cfg in
c4s) timeout 900 python scripts/prof_traffic.py r02_c4shard_1250k_chunked --aggregates 1250000 > gpurun_out/prof_c4s.log 2>&1; tail -1 gpurun_out/prof_c4s.log;;
c2) timeout 600 python scripts/prof_traffic.py r02_c2_rows --workload c2 > gpurun_out/prof_c2.log 2>&1; tail -1 gpurun_out/prof_c2.log;;
c3) PROF_SKIP_SQ=1 timeout 700 python scripts/prof_traffic.py r02_c3_10Magg_sorted > gpurun_out/prof_c3.log 2>&1; tail -1 gpurun_out/prof_c3.log;;
300k) PROF_SKIP_SQ=1 timeout 500 python scripts/prof_traffic.py r02_zipf_300k_chunked --aggregates 300000 > gpurun_out/prof_300k.log 2>&1; tail -1 gpurun_out/prof_300k.log;;
100k) PROF_SKIP_SQ=1 timeout 500 python scripts/prof_traffic.py r02_zipf_100k_flat --aggregates 100000 > gpurun_out/prof_100k.log 2>&1; tail -1 gpurun_out/prof_100k.log;;
esac
done
