mkdir -p gpurun_out/r3e
timeout 600 python scripts/experiments/stream_data_probe.py > gpurun_out/r3e/data_probe.log 2>&1
cat gpurun_out/r3e/data_probe.log
