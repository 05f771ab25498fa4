mkdir -p gpurun_out/r3v
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v/pytest.log
grep -v amdgpu.ids gpurun_out/r3v/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
cd /tmp && export TMPDIR=/tmp
PROF_PASS_TIMEOUT=200 timeout 1100 python $GRAFT_REPO_ROOT/scripts/prof_traffic.py r03_v2_ledger_2Magg --workload v2 > $GRAFT_REPO_ROOT/gpurun_out/r3v/prof.log 2>&1; echo "prof rc=$?"
tail -30 $GRAFT_REPO_ROOT/gpurun_out/r3v/prof.log
