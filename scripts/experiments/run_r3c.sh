mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or chunked_rows or random_log or zipf_csr or fixed_fan or golden" > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
tail -3 gpurun_out/r3c/pytest.log
SHAPES=c3,c4s,c2 CONFIGS=2:8,2:9,2:6,2:4,1:16,1:12,1:8 FOLDS=30 timeout 900 python scripts/experiments/tiled_probe.py > gpurun_out/r3c/probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r3c/probe.log
cat gpurun_out/r3c/probe.log
