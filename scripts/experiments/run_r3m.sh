mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_bench_rehearsal.py -x -q -m gpu -s > gpurun_out/r3m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3m/pytest.log
grep -v "amdgpu.ids" gpurun_out/r3m/pytest.log | tail -25
