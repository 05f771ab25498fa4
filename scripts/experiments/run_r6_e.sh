export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "counting_sort or chunked_rows or random_log_shapes or tile_major or ragged" 2>&1 | tail -5
timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --parity none > gpurun_out/r06_bench_c3_quick.json 2> gpurun_out/r06_bench_c3_quick.err; tail -2 gpurun_out/r06_bench_c3_quick.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_c3_quick.json').read().strip().splitlines()[-1])
print(json.dumps(d['roofline'])[:1200]); print(json.dumps(d['one_shot'])[:600])
PY
timeout 600 python bench.py --workload c4-shard --steps 50 > gpurun_out/r06_bench_c4s_quick.json 2> gpurun_out/r06_bench_c4s_quick.err; tail -2 gpurun_out/r06_bench_c4s_quick.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_c4s_quick.json').read().strip().splitlines()[-1])
print(json.dumps(d['roofline'])[:600]); print(json.dumps(d['one_shot'])[:600], d['cpu_baseline'])
PY
