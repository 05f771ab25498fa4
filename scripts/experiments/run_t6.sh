mkdir -p gpurun_out/t6
timeout 600 python -m pytest tests/test_slots.py -x -q -m gpu -s > gpurun_out/t6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t6/pytest.log; tail -5 gpurun_out/t6/pytest.log
timeout 600 python scripts/experiments/slots_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t6/slots.log
