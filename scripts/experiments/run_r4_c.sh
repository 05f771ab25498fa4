mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_ingest_gpu.py tests/test_store.py tests/test_abi.py -x -q -m gpu > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4c/pytest.log
PROF_PASSES=trace timeout 300 python scripts/prof_ingest.py r04_e2e_sections --workload e2e --steps 6 > gpurun_out/r4c/prof.log 2>&1; echo "prof rc=$?"
head -24 gpurun_out/prof_r04_e2e_sections/r04_e2e_sections_summary.txt
python -c "
import json; d=json.load(open('gpurun_out/prof_r04_e2e_sections/r04_e2e_sections_bench.json')); c=d['config']; print(d['value'], c['fetch_ms'], c['host_framing_ms_per_fetch'], c['device_decode_groupby_fold_ms_per_fetch'], d['cpu_baseline'])"
