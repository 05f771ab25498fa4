mkdir -p gpurun_out/r3aa
timeout 900 python -m pytest tests/test_store.py tests/test_event_decode.py tests/test_ingest_gpu.py tests/test_persistence.py -x -q -m gpu -s > gpurun_out/r3aa/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3aa/pytest.log
grep -v amdgpu.ids gpurun_out/r3aa/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30
