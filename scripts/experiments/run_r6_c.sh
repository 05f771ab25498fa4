export TMPDIR=/tmp
mkdir -p gpurun_out
export SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_exp.so
SHAPE=c3 ROUNDS=3 FOLDS=5 SCHEMAS=builtin,counter VARIANTS="builtin:aot:16,builtin:rtc:16,builtin:rtc:16:::SURGE_EXP_WAIT_BEFORE_WALK=1,builtin:rtc:16:::SURGE_EXP_SKIP_APPLY=1,builtin:rtc:16:::SURGE_EXP_SKIP_APPLY=1+SURGE_EXP_WAIT_BEFORE_WALK=1,counter:aot:16,counter:rtc:16:::SURGE_EXP_WAIT_BEFORE_WALK=1,builtin:rtc:16:6::SURGE_EXP_WAIT_BEFORE_WALK=1,builtin:rtc:16:6::SURGE_EXP_SKIP_APPLY=1,builtin:rtc:8:::SURGE_EXP_SKIP_APPLY=1,builtin:rtc:8:::SURGE_EXP_WAIT_BEFORE_WALK=1" timeout 900 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab3_c3.jsonl 2> gpurun_out/r06_lane_spec_ab3_c3.err; tail -3 gpurun_out/r06_lane_spec_ab3_c3.err
python - <<'PY'
import json
for l in open('gpurun_out/r06_lane_spec_ab3_c3.jsonl'):
    d=json.loads(l); print(d['schema'],d['build'],d['lane_events'],d['waves_per_cu'],d['extra'],round(d['median_ms'],3),round(d['frac_of_8TBps'],4),d['states_equal_first_variant'])
PY
