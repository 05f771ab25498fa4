mkdir -p gpurun_out/r3z
timeout 600 python scripts/ingest_gpu_bench.py 8000 > gpurun_out/r3z/ingest_gpu.json 2> gpurun_out/r3z/ingest_gpu.err; echo "rc=$?"
cat gpurun_out/r3z/ingest_gpu.json; tail -3 gpurun_out/r3z/ingest_gpu.err
