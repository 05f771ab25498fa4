export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 300 > /tmp/suite.log 2>&1
grep -E "passed|failed|error" /tmp/suite.log | tail -5 | tee gpurun_out/r06_gpu_suite.txt
grep -E "^FAILED|^ERROR" /tmp/suite.log | head -10 | tee -a gpurun_out/r06_gpu_suite.txt
