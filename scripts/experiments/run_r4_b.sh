mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4b/pytest.log
PROF_PASSES=trace timeout 300 python scripts/prof_ingest.py r04_e2e_lz4_two_pass --workload e2e --steps 6 > gpurun_out/r4b/prof.log 2>&1; echo "prof rc=$?"
head -24 gpurun_out/prof_r04_e2e_lz4_two_pass/r04_e2e_lz4_two_pass_summary.txt
python -c "
import json; d=json.load(open('gpurun_out/prof_r04_e2e_lz4_two_pass/r04_e2e_lz4_two_pass_bench.json')); c=d['config']; print(d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['device_decode_groupby_fold_ms_per_fetch'], d['cpu_baseline'])"
