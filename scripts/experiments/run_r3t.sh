mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_event_decode.py tests/test_store.py tests/test_persistence.py tests/test_abi.py tests/test_comm.py -x -q -m gpu -s > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3t/pytest.log
grep -v amdgpu.ids gpurun_out/r3t/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
