mkdir -p gpurun_out/r3r
for sh in c3 z4m c4s; do for w in 6 8; do VARIANTS=main,prev SHAPE=$sh ALGO=7 SUBS=2 WAVES=$w ROUNDS=4 FOLDS=15 timeout 400 python scripts/experiments/ab_seq.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3r/ab.log; done; done
cat gpurun_out/r3r/ab.log
