mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_slots.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3
ZIPF_AGGS=1250000 FOLDS=60 timeout 300 python scripts/experiments/variance_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
ZIPF_AGGS=10000000 FOLDS=12 timeout 300 python scripts/experiments/variance_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
