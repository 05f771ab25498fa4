export TMPDIR=/tmp
mkdir -p gpurun_out
export SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_exp.so
V=""
for w in 1 4 6 8; do V="$V,builtin:aot:16:$w,builtin:rtc:16:$w,builtin:rtc:16:$w::SURGE_EXP_SKIP_APPLY=1,builtin:rtc:8:$w"; done
V="$V,builtin:rtc:16:5,builtin:rtc:16:7,builtin:rtc:8:12,builtin:rtc:8:16,builtin:aot:8:12,counter:aot:16:8,counter:rtc:16:8,counter:rtc:16:6,counter:rtc:8:12,counter:rtc:8:16"
SHAPE=c3 ROUNDS=2 FOLDS=3 SCHEMAS=builtin,counter VARIANTS="${V#,}" timeout 900 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab4_c3.jsonl 2> gpurun_out/r06_lane_spec_ab4_c3.err; tail -3 gpurun_out/r06_lane_spec_ab4_c3.err
python - <<'PY'
import json
for l in open('gpurun_out/r06_lane_spec_ab4_c3.jsonl'):
    d=json.loads(l)
    w=int(d['waves_per_cu']); le=d['lane_events']
    T=w*le*1024*256/(74.4e9/(d['median_ms']*1e-3))*1e6
    print(d['schema'],d['build'],le,'waves',w,d['extra'][:20],'ms %.2f'%d['median_ms'],'frac %.3f'%d['frac_of_8TBps'],'T_per_tile_us %.2f'%T, d['states_equal_first_variant'])
PY
