mkdir -p gpurun_out/subset
timeout 300 python -m pytest tests/test_ingest_gpu.py tests/test_frame_gpu.py tests/test_store.py tests/test_event_decode.py -q -m gpu > gpurun_out/subset/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/subset/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/subset/pytest.log | tail -10
