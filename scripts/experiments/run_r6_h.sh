export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu --timeout 120 -k "crc32c" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 120 2>&1 | tail -12
