mkdir -p gpurun_out/t4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_store.py tests/test_persistence.py -x -q -m gpu -k "batch or append or micro or stream or store or persist or new_aggregates or bulk or commands or multilanguage or publish or exceptions or properly" > gpurun_out/t4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4/pytest.log
tail -6 gpurun_out/t4/pytest.log
timeout 300 python scripts/stream_bench.py --aggregates 200000 --batches 12 --batch-events 20000 --snapshot-every 4 --verify > gpurun_out/t4/stream_verify.json 2> gpurun_out/t4/stream_verify.err; tail -2 gpurun_out/t4/stream_verify.err; cat gpurun_out/t4/stream_verify.json
timeout 900 python scripts/stream_bench.py > gpurun_out/t4/stream_c5.json 2> gpurun_out/t4/stream_c5.err; tail -2 gpurun_out/t4/stream_c5.err; cat gpurun_out/t4/stream_c5.json
timeout 900 python scripts/stream_bench.py --batch-events 1000000 --batches 60 --snapshot-every 3 > gpurun_out/t4/stream_c5_1m.json 2> gpurun_out/t4/stream_c5_1m.err; cat gpurun_out/t4/stream_c5_1m.json
