mkdir -p gpurun_out/r2m
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
tail -6 gpurun_out/r2m/pytest.log
ZIPF_SIZES=150000,300000,800000,1250000,3000000 VARIANTS=0:0:0,2:0:0,4:0:0,5:0:16 timeout 900 python scripts/experiments/chunk_sweep.py > gpurun_out/r2m/sweep.log 2>&1
cat gpurun_out/r2m/sweep.log
