mkdir -p gpurun_out/r3ai
timeout 600 python -m pytest tests/test_store.py -x -q -m gpu > gpurun_out/r3ai/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3ai/pytest.log
grep -v amdgpu.ids gpurun_out/r3ai/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
timeout 600 python bench.py --workload c5 > gpurun_out/r3ai/bench_c5.json 2> gpurun_out/r3ai/bench_c5.err; echo "c5 rc=$?"; tail -3 gpurun_out/r3ai/bench_c5.err; python -c "
import json; d=json.load(open('gpurun_out/r3ai/bench_c5.json')); c=d['config']; print('c5', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['snapshot_framing_ms_in_background'], d['cpu_baseline']['gpu_matches_cpu_full_run'])"
timeout 600 python bench.py --workload c5 --device-batches > gpurun_out/r3ai/bench_c5_dev.json 2> gpurun_out/r3ai/bench_c5_dev.err; echo "c5 dev rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3ai/bench_c5_dev.json')); c=d['config']; print('c5 dev', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['snapshot_framing_ms_in_background'], d['cpu_baseline']['gpu_matches_cpu_full_run'])"
