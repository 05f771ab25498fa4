mkdir -p gpurun_out/r4m
timeout 600 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu > gpurun_out/r4m/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r4m/pytest.log | cut -c1-300
PROF_PASSES=trace PROF_PASS_TIMEOUT=600 timeout 900 python scripts/prof_ingest.py r04_e2e_serial --workload e2e --events-cap 2 --serial-framing > gpurun_out/r4m/prof.log 2>&1; echo "prof rc=$?"
head -14 gpurun_out/prof_r04_e2e_serial/r04_e2e_serial_summary.txt
