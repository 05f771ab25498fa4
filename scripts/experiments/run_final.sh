mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -q -m gpu -rs > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
tail -8 gpurun_out/final/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; echo "bench rc=$?"; tail -2 gpurun_out/final/bench_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_n1.json'))
r=d['roofline']; s=d['secondary']['roofline']
print('value',d['value'],'ms/step',d['ms_per_step'],'frac',r['frac'],'traffic',r['traffic'],r['kernel_ms_min_median_max'])
print('secondary',d['secondary']['value'],s['frac'],s['traffic'],s['kernel_ms_min_median_max'])
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['gpu_matches_cpu_on_sample'],d['cpu_baseline']['thread_scaling_efficiency'])
PY
