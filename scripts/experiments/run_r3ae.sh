mkdir -p gpurun_out/r3ae
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3ae/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3ae/pytest.log
grep -v amdgpu.ids gpurun_out/r3ae/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
timeout 600 python bench.py --workload e2e > gpurun_out/r3ae/bench_e2e.json 2> gpurun_out/r3ae/bench_e2e.err; echo "e2e rc=$?"; tail -c 2300 gpurun_out/r3ae/bench_e2e.json; tail -3 gpurun_out/r3ae/bench_e2e.err
timeout 600 python bench.py --workload e2e --codec none > gpurun_out/r3ae/bench_e2e_none.json 2> gpurun_out/r3ae/bench_e2e_none.err; echo "e2e none rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3ae/bench_e2e_none.json')); print('e2e none', d['value'], d['config']['fetch_ms'], d['config']['host_framing_ms_per_fetch'], d['config']['device_decode_groupby_fold_ms_per_fetch'])"
cd /tmp && export TMPDIR=/tmp
PROF_PASS_TIMEOUT=240 PROF_SKIP_SQ=1 timeout 1000 python $GRAFT_REPO_ROOT/scripts/prof_traffic.py r03_c3_10Magg_tiled_b > $GRAFT_REPO_ROOT/gpurun_out/r3ae/prof_c3.log 2>&1; echo "prof c3 rc=$?"; tail -4 $GRAFT_REPO_ROOT/gpurun_out/r3ae/prof_c3.log
