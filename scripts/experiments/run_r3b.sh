mkdir -p gpurun_out/r3b
for v in skip fixtab; do
  echo "== variant $v" >> gpurun_out/r3b/probe.log
  SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_$v.so NOCHECK=1 SHAPES=c3,c4s CONFIGS=2:8,1:12,2:4,1:8 FOLDS=20 timeout 600 python scripts/experiments/tiled_probe.py >> gpurun_out/r3b/probe.log 2>&1
done
cat gpurun_out/r3b/probe.log
