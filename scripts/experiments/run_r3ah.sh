mkdir -p gpurun_out/r3ah
timeout 600 python -m pytest tests/test_store.py tests/test_persistence.py tests/test_properties.py -x -q -m gpu > gpurun_out/r3ah/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3ah/pytest.log
grep -v amdgpu.ids gpurun_out/r3ah/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
timeout 600 python bench.py --workload c5 > gpurun_out/r3ah/bench_c5.json 2> gpurun_out/r3ah/bench_c5.err; echo "c5 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3ah/bench_c5.json')); c=d['config']; print('c5', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['ingest_only_events_per_sec_synced_per_batch'], d['cpu_baseline']['gpu_matches_cpu_full_run'])"
