mkdir -p gpurun_out/framer
timeout 600 python -m pytest tests/test_frame_gpu.py tests/test_store.py -q -m gpu -x > gpurun_out/framer/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/framer/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/framer/pytest.log | tail -15
for mode in device host; do
  flag=""; [ $mode = host ] && flag="--host-framing"
  timeout 400 python bench.py --workload c5 $flag > gpurun_out/framer/c5_$mode.json 2> gpurun_out/framer/c5_$mode.err; echo "c5 $mode rc=$?"; tail -2 gpurun_out/framer/c5_$mode.err | grep -v amdgpu
  python -c "
import json; d=json.load(open('gpurun_out/framer/c5_$mode.json')); c=d['config']; print('c5 $mode', d['value'], c['snapshot_ms'], c['snapshot_parts_ms_mean'], c['snapshot_record_batch_bytes_mean'], d['cpu_baseline'].get('gpu_matches_cpu') if isinstance(d.get('cpu_baseline'),dict) else None)"
done
