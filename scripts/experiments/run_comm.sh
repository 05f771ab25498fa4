mkdir -p gpurun_out/comm
timeout 900 python -m pytest tests/test_comm.py -x -q -m gpu -rs > gpurun_out/comm/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/comm/pytest.log
tail -25 gpurun_out/comm/pytest.log
