export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 200 2>&1 | tail -4
timeout 600 python -m pytest tests/test_bench_rehearsal.py -x -q -m gpu -k "e2e" --timeout 300 2>&1 | tail -3
bash scripts/experiments/run_r6_x.sh
