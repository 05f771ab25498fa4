mkdir -p gpurun_out/r3y
timeout 300 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu > gpurun_out/r3y/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3y/pytest.log
grep -v amdgpu.ids gpurun_out/r3y/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40
