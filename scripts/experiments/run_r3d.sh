mkdir -p gpurun_out/r3d
timeout 300 python scripts/experiments/dbg_flat.py > gpurun_out/r3d/dbg.log 2>&1
cat gpurun_out/r3d/dbg.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log
tail -15 gpurun_out/r3d/pytest.log
