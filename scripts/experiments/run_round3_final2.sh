# Re-verification after the device framer / single-pass LZ4 / LDS-staged resolve (no fold kernel source changed since
# run_round3_final.sh: the profiles of that run stand).   gpurun -- bash scripts/experiments/run_round3_final2.sh
mkdir -p gpurun_out/r3final2
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3final2/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3final2/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|pytest rc" gpurun_out/r3final2/pytest.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final2/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3final2/smoke.log
timeout 600 python bench.py > gpurun_out/r3final2/bench_n1.json 2> gpurun_out/r3final2/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3final2/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_matches_cpu_full_log'))"
timeout 300 python bench.py --workload c5 --device-batches > gpurun_out/r3final2/c5_device_batches.json 2> gpurun_out/r3final2/c5_device_batches.err; echo "c5 device-batches rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3final2/c5_device_batches.json')); c=d['config']; print('c5 dev', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['snapshot_parts_ms_mean'], d['cpu_baseline'].get('gpu_matches_cpu_full_run'))"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3final2/c5_trace
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3final2/c5_trace -o c5 -- python $R/bench.py --workload c5 --steps 120 --no-cpu-baseline > $R/gpurun_out/r3final2/c5_trace.log 2>&1; echo "c5 trace rc=$?"
