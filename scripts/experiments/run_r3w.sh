mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_store.py tests/test_persistence.py -x -q -m gpu -k "json or bank or encoder or snapshot or protobuf" > gpurun_out/r3w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3w/pytest.log
grep -v amdgpu.ids gpurun_out/r3w/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
