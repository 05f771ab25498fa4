mkdir -p gpurun_out/framer
timeout 600 python -m pytest tests/test_frame_gpu.py tests/test_store.py -q -m gpu > gpurun_out/framer/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/framer/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/framer/pytest.log | tail -15
