mkdir -p gpurun_out/r3x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3x/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3x/pytest.log
grep -v amdgpu.ids gpurun_out/r3x/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err; echo "bench rc=$?"
tail -c 4500 gpurun_out/r3x/bench.json; tail -3 gpurun_out/r3x/bench.err
