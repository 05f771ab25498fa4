# full verification + the profiles the traffic manifest is built from (one gpurun call); every step bounded
mkdir -p gpurun_out/final
rm -rf gpurun_out/prof_r02_*
timeout 900 python -m pytest tests -q -m gpu -rs > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
grep -E "passed|failed|error|rc=" gpurun_out/final/pytest.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; echo "bench rc=$?"
ZIPF_AGGS=1250000 FOLDS=40 timeout 200 python scripts/experiments/variance_probe.py 2>&1 | grep "^rep" 
export PROF_PASS_TIMEOUT=200
timeout 900 python scripts/prof_traffic.py r02_c4shard_1250k_chunked --aggregates 1250000 > gpurun_out/prof_c4s.log 2>&1; tail -1 gpurun_out/prof_c4s.log
timeout 600 python scripts/prof_traffic.py r02_c2_rows --workload c2 > gpurun_out/prof_c2.log 2>&1; tail -1 gpurun_out/prof_c2.log
PROF_SKIP_SQ=1 timeout 700 python scripts/prof_traffic.py r02_c3_10Magg_sorted > gpurun_out/prof_c3.log 2>&1; tail -1 gpurun_out/prof_c3.log
PROF_SKIP_SQ=1 timeout 500 python scripts/prof_traffic.py r02_zipf_300k_chunked --aggregates 300000 > gpurun_out/prof_300k.log 2>&1; tail -1 gpurun_out/prof_300k.log
PROF_SKIP_SQ=1 timeout 500 python scripts/prof_traffic.py r02_zipf_100k_flat --aggregates 100000 > gpurun_out/prof_100k.log 2>&1; tail -1 gpurun_out/prof_100k.log
# optional refreshes (host-side encoders changed since the committed numbers)
timeout 300 python scripts/snapshot_bench.py 10000000 2>/dev/null | tail -1 > gpurun_out/final/snapshot_n2_10M.json; cut -c1-300 gpurun_out/final/snapshot_n2_10M.json
timeout 300 python scripts/stream_bench.py 2>/dev/null | tail -1 > gpurun_out/final/stream_c5.json; cut -c1-400 gpurun_out/final/stream_c5.json
