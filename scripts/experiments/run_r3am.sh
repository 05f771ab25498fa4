mkdir -p gpurun_out/r3am
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3am/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3am/pytest.log
grep -v amdgpu.ids gpurun_out/r3am/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
for mode in overlap serial; do
  flag=""; [ $mode = serial ] && flag="--serial-framing"
  timeout 600 python bench.py --workload e2e $flag > gpurun_out/r3am/bench_e2e_$mode.json 2> gpurun_out/r3am/bench_e2e_$mode.err; echo "e2e $mode rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r3am/bench_e2e_$mode.json')); print('e2e $mode', d['value'], d['config']['fetch_ms'], d['config']['host_framing_ms_per_fetch'], d['config']['device_decode_groupby_fold_ms_per_fetch'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_decoded_events'])"
done
