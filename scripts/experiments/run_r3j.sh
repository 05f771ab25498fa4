mkdir -p gpurun_out/r3j
SURGE_DBG_PRINT=1 MODES=0 HANDLES=7 SHAPE=z4m SUBS=2 WAVES=6 ROUNDS=3 FOLDS=10 timeout 600 python scripts/experiments/ab_modes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3j/ab.log
cat gpurun_out/r3j/ab.log
