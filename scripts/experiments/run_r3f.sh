mkdir -p gpurun_out/r3f
for v in skip skiplin skipnost lin; do
  echo "== variant $v" >> gpurun_out/r3f/probe.log
  SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_$v.so NOCHECK=1 SHAPES=c3 CONFIGS=2:8,2:6,1:8 FOLDS=20 timeout 600 python scripts/experiments/tiled_probe.py >> gpurun_out/r3f/probe.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r3f/probe.log
