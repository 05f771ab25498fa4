export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -8 | tee gpurun_out/r06_gpu_suite.txt
