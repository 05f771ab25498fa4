# The last verification run of round 3 on an MI355X box: full GPU suite, smoke(), the four rocprofv3 profiles the
# traffic manifest is built from, a kernel trace of the e2e workload and the default bench line.   gpurun -- bash scripts/experiments/run_round3_final.sh
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3final/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3final/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|pytest rc" gpurun_out/r3final/pytest.log | tail -12
if [ $rc -ne 0 ]; then grep -v amdgpu.ids gpurun_out/r3final/pytest.log | tail -60; exit 1; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3final/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PROF_PASS_TIMEOUT=200 timeout 1300 python $R/scripts/prof_traffic.py r03_c3_10Magg_tiled > $R/gpurun_out/r3final/prof_c3.log 2>&1; echo "prof c3 rc=$?"; tail -2 $R/gpurun_out/r3final/prof_c3.log
PROF_PASS_TIMEOUT=120 timeout 700 python $R/scripts/prof_traffic.py r03_c4shard_1250k_tiled --aggregates 1250000 --steps 100 > $R/gpurun_out/r3final/prof_c4s.log 2>&1; echo "prof c4s rc=$?"; tail -2 $R/gpurun_out/r3final/prof_c4s.log
PROF_PASS_TIMEOUT=120 timeout 700 python $R/scripts/prof_traffic.py r03_v2_ledger_2Magg --workload v2 > $R/gpurun_out/r3final/prof_v2.log 2>&1; echo "prof v2 rc=$?"; tail -2 $R/gpurun_out/r3final/prof_v2.log
PROF_PASS_TIMEOUT=120 timeout 500 python $R/scripts/prof_traffic.py r03_zipf_100k_flat --aggregates 100000 --algo flat > $R/gpurun_out/r3final/prof_flat.log 2>&1; echo "prof flat rc=$?"; tail -2 $R/gpurun_out/r3final/prof_flat.log
mkdir -p $R/gpurun_out/r3final/e2e_trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3final/e2e_trace -o e2e -- python $R/bench.py --workload e2e --steps 6 > $R/gpurun_out/r3final/e2e_trace.log 2>&1; echo "e2e trace rc=$?"
cd $R
python scripts/merge_manifest.py r03_c3_10Magg_tiled r03_c4shard_1250k_tiled r03_v2_ledger_2Magg r03_zipf_100k_flat
timeout 600 python bench.py > gpurun_out/r3final/bench_n1.json 2> gpurun_out/r3final/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3final/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source','')[:60], d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_matches_cpu_full_log'))"
