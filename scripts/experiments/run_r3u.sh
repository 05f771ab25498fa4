mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_slots.py tests/test_abi.py -x -q -m gpu -s > gpurun_out/r3u/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3u/pytest.log
grep -v amdgpu.ids gpurun_out/r3u/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
timeout 600 python bench.py --workload v2 --steps 10 --warmup 2 > gpurun_out/r3u/bench_v2.json 2> gpurun_out/r3u/bench_v2.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3u/bench_v2.json; tail -5 gpurun_out/r3u/bench_v2.err
