mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_store.py tests/test_slots.py tests/test_persistence.py -x -q -m gpu > gpurun_out/r3s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3s/pytest.log
grep -v amdgpu.ids gpurun_out/r3s/pytest.log | tail -15
timeout 900 python bench.py --workload c5 > gpurun_out/r3s/c5.json 2> gpurun_out/r3s/c5.err; echo "c5 rc=$?"; tail -3 gpurun_out/r3s/c5.err; cat gpurun_out/r3s/c5.json
timeout 600 python bench.py --workload c5 --device-batches --no-cpu-baseline > gpurun_out/r3s/c5_dev.json 2> gpurun_out/r3s/c5_dev.err; echo "c5dev rc=$?"; cat gpurun_out/r3s/c5_dev.json
