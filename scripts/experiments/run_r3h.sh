mkdir -p gpurun_out/r3h
for cfg in "2 6" "2 8" "1 8"; do set -- $cfg
VARIANTS=main,old,lin SHAPE=z4m SUBS=$1 WAVES=$2 ROUNDS=8 FOLDS=10 timeout 600 python scripts/experiments/ab_inproc.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3h/ab.log
done
cat gpurun_out/r3h/ab.log
