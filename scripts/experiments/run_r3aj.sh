mkdir -p gpurun_out/r3aj
timeout 600 python scripts/ingest_gpu_mt_bench.py 6 > gpurun_out/r3aj/ingest_mt.json 2> gpurun_out/r3aj/ingest_mt.err; echo "rc=$?"; grep -v amdgpu gpurun_out/r3aj/ingest_mt.err | tail -8; cat gpurun_out/r3aj/ingest_mt.json
