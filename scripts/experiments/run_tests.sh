mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests/pytest.log
tail -12 gpurun_out/tests/pytest.log
