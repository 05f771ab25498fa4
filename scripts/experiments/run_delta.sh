mkdir -p gpurun_out/delta
timeout 900 python -m pytest tests/test_store.py tests/test_frame_gpu.py tests/test_gpu_parity.py tests/test_persistence.py tests/test_slots.py -q -m gpu > gpurun_out/delta/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/delta/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/delta/pytest.log | tail -8
timeout 400 python bench.py --workload c5 > gpurun_out/delta/c5.json 2> gpurun_out/delta/c5.err; echo "c5 rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/delta/c5.json')); c=d['config']; print('c5', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['snapshot_parts_ms_mean'], d['cpu_baseline'].get('gpu_matches_cpu_full_run'))"
