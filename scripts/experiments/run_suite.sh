mkdir -p gpurun_out/suite
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/suite/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/suite/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|pytest rc" gpurun_out/suite/pytest.log | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/suite/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/suite/smoke.log
