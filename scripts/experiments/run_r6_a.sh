set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane_kernels or random_log_shapes or flat_kernel_compiled or chunked_rows or golden or auto_picks" 2>&1 | tail -15
SHAPE=c3 ROUNDS=4 FOLDS=5 timeout 900 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab_c3.jsonl 2> gpurun_out/r06_lane_spec_ab_c3.err; tail -3 gpurun_out/r06_lane_spec_ab_c3.err
cat gpurun_out/r06_lane_spec_ab_c3.jsonl | cut -c1-330
SHAPE=c4s ROUNDS=6 FOLDS=10 timeout 600 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab_c4s.jsonl 2>&1
cat gpurun_out/r06_lane_spec_ab_c4s.jsonl | cut -c1-330
SHAPE=c2 ROUNDS=6 FOLDS=10 timeout 600 python scripts/lane_spec_ab.py > gpurun_out/r06_lane_spec_ab_c2.jsonl 2>&1
cat gpurun_out/r06_lane_spec_ab_c2.jsonl | cut -c1-330
