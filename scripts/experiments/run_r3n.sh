mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3n/pytest.log
tail -4 gpurun_out/r3n/pytest.log
for sh in z100k z300k c4s; do VARIANTS=main,prev SHAPE=$sh ALGO=2 ROUNDS=6 FOLDS=20 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3n/ab.log; done
VARIANTS=main,prev SHAPE=c2 ALGO=3 ROUNDS=6 FOLDS=20 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3n/ab.log
VARIANTS=main,prev SHAPE=c2 ALGO=1 ROUNDS=6 FOLDS=20 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3n/ab.log
VARIANTS=main,prev SHAPE=c4s ALGO=5 ROUNDS=6 FOLDS=20 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3n/ab.log
cat gpurun_out/r3n/ab.log
