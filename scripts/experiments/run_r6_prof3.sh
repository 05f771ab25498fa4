export TMPDIR=/tmp
PROF_PASSES=trace,sq1,sq2 PROF_PASS_TIMEOUT=200 timeout 900 python scripts/prof_ingest.py r06_e2e_final --workload e2e --steps 12 --warmup 2 --parity none 2>&1 | tail -40
rm -rf gpurun_out/prof_r06_e2e_final/trace gpurun_out/prof_r06_e2e_final/sq1 gpurun_out/prof_r06_e2e_final/sq2
ls -la gpurun_out/prof_r06_e2e_final | head
