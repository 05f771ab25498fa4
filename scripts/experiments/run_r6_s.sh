export TMPDIR=/tmp
mkdir -p gpurun_out
export SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_exp.so
SHAPE=c3 ROUNDS=2 FOLDS=3 SCHEMAS=builtin VARIANTS="builtin:aot:16:8,builtin:rtc:32:4,builtin:rtc:32:3,builtin:rtc:32:2,builtin:rtc:32:4:::SURGE_EXP_SKIP_APPLY=1" timeout 900 python scripts/lane_spec_ab.py 2> gpurun_out/x.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['shape'],d['schema'],d['build'],d['lane_events'],'waves',d['waves_per_cu'],d['extra'][:20],'ms %.3f'%d['median_ms'],'frac %.4f'%d['frac_of_8TBps'],d['states_equal_first_variant'])
"
tail -2 gpurun_out/x.err
for shape in c4s c2; do
SHAPE=$shape ROUNDS=4 FOLDS=10 timeout 600 python scripts/lane_spec_ab.py 2> gpurun_out/x.err | tee gpurun_out/r06_lane_spec_ab5_$shape.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['shape'],d['schema'],d['build'],d['lane_events'],'waves',d['waves_per_cu'],d['extra'][:20],'ms %.4f'%d['median_ms'],'frac %.4f'%d['frac_of_8TBps'],d['states_equal_first_variant'])
"
done
