mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3q/pytest.log
tail -4 gpurun_out/r3q/pytest.log
for sh in z4m c4s c2; do for w in 6 8; do VARIANTS=main,prev SHAPE=$sh ALGO=7 SUBS=2 WAVES=$w ROUNDS=6 FOLDS=10 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3q/ab.log; done; done
VARIANTS=main,prev SHAPE=z4m ALGO=4 ROUNDS=4 FOLDS=10 timeout 300 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3q/ab.log
cat gpurun_out/r3q/ab.log
