mkdir -p gpurun_out/subset
timeout 500 python -m pytest tests/test_bench_rehearsal.py tests/test_frame_gpu.py -q -m gpu -k "c5 or framer" > gpurun_out/subset/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/subset/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/subset/pytest.log | tail -10
