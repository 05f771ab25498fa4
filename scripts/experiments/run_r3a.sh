mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or chunked_rows or random_log or zipf_csr or csr_window or auto_picks" > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
SHAPES=c4s,c2,c3 timeout 900 python scripts/experiments/tiled_probe.py > gpurun_out/r3a/probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r3a/probe.log
cat gpurun_out/r3a/probe.log | tail -40
