# The round's closing run: full GPU suite, smoke(), default bench line, the AUTO policy sweep on the final sources.
mkdir -p gpurun_out/r3final3
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3final3/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3final3/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|pytest rc" gpurun_out/r3final3/pytest.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final3/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3final3/smoke.log
timeout 600 python bench.py > gpurun_out/r3final3/bench_n1.json 2> gpurun_out/r3final3/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3final3/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_matches_cpu_full_log'))"
timeout 400 python scripts/auto_policy_sweep.py > gpurun_out/r3final3/auto_policy_sweep.jsonl 2> gpurun_out/r3final3/auto_policy_sweep.err; echo "sweep rc=$?"
python - <<'P'
import json
for line in open('gpurun_out/r3final3/auto_policy_sweep.jsonl'):
    r=json.loads(line); print(r['shape'], r['aggregates'], 'auto', r['auto'], {k: round(v['frac'],3) for k,v in r.items() if isinstance(v,dict) and 'frac' in v})
P
