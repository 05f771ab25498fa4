export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_store.py tests/test_ingest_gpu.py -x -q -m gpu 2>&1 | tail -8
