mkdir -p gpurun_out/t5
timeout 900 python -m pytest tests/test_slots.py -x -q -m gpu -s > gpurun_out/t5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t5/pytest.log
tail -25 gpurun_out/t5/pytest.log
