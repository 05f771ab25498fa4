mkdir -p gpurun_out/t3
timeout 1500 python -m pytest tests/test_persistence.py tests/test_store.py tests/test_abi.py tests/test_comm.py -x -q -m gpu -rs > gpurun_out/t3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t3/pytest.log
tail -15 gpurun_out/t3/pytest.log
timeout 600 python scripts/snapshot_bench.py 2000000 > gpurun_out/t3/snapshot.json 2> gpurun_out/t3/snapshot.err; tail -2 gpurun_out/t3/snapshot.err; cat gpurun_out/t3/snapshot.json
