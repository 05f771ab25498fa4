set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
SHAPE=c3 ROUNDS=3 FOLDS=5 SCHEMAS=builtin,counter VARIANTS="builtin:aot:16,builtin:rtc:16::1552,builtin:rtc:16:9,builtin:rtc:16:12,builtin:rtc:8:16,builtin:rtc:8:12,builtin:rtc:32:4,builtin:rtc:32:8,counter:aot:16,counter:rtc:16::1552,counter:rtc:32:4,counter:rtc:8:16,builtin:rtc:16:6:1552,builtin:aot:16:6" timeout 900 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab2_c3.jsonl 2> gpurun_out/r06_lane_spec_ab2_c3.err; tail -3 gpurun_out/r06_lane_spec_ab2_c3.err
cut -c1-300 gpurun_out/r06_lane_spec_ab2_c3.jsonl
