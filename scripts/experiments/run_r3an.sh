mkdir -p gpurun_out/r3an
timeout 300 python -m pytest tests/test_store.py tests/test_bench_rehearsal.py -q -m gpu -k "fetch or e2e" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 400 python scripts/flat_small_sweep.py > gpurun_out/r3an/flat_sweep.jsonl 2> gpurun_out/r3an/flat_sweep.err; echo "sweep rc=$?"
python - <<'P'
import json
for line in open('gpurun_out/r3an/flat_sweep.jsonl'):
    r=json.loads(line); print(r['shape'], r['aggregates'])
    for k,v in r.items():
        if isinstance(v,dict): print('   ', k, v)
P
for mode in overlap; do
  timeout 600 python bench.py --workload e2e > gpurun_out/r3an/bench_e2e_$mode.json 2> gpurun_out/r3an/bench_e2e_$mode.err; echo "e2e $mode rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r3an/bench_e2e_$mode.json')); print('e2e $mode', d['value'], d['config']['fetch_ms'], d['config']['host_framing_ms_per_fetch'], d['config']['device_decode_groupby_fold_ms_per_fetch'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_decoded_events'])"
done
PROF_PASS_TIMEOUT=120 timeout 700 python scripts/prof_traffic.py r03_zipf_100k_flat --aggregates 100000 --algo flat 2>&1 | grep -v amdgpu.ids | tail -4
