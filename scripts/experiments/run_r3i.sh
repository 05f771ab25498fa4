mkdir -p gpurun_out/r3i
for cfg in "2 6" "2 8" "1 8"; do set -- $cfg
MODES=0,3,1 HANDLES=3 SHAPE=z4m SUBS=$1 WAVES=$2 ROUNDS=5 FOLDS=10 timeout 600 python scripts/experiments/ab_modes.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3i/ab.log
done
cat gpurun_out/r3i/ab.log
