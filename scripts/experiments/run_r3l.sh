mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_bench_rehearsal.py -x -q -m gpu -s > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l/pytest.log
grep -v "amdgpu.ids" gpurun_out/r3l/pytest.log | tail -25
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3l/bench_n1.json 2> gpurun_out/r3l/bench_n1.err; echo "bench rc=$?"; tail -3 gpurun_out/r3l/bench_n1.err
cat gpurun_out/r3l/bench_n1.json
