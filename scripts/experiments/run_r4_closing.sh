# round 4, closing verification on the final sources: the whole GPU suite, smoke(), the default line, e2e (default and twice the events)
mkdir -p gpurun_out/r4closing
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r4closing/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r4closing/pytest.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4closing/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4closing/smoke.log
timeout 600 python bench.py --workload e2e > gpurun_out/r4closing/e2e.json 2>/dev/null; echo "e2e rc=$?"
timeout 900 python bench.py --workload e2e --events-cap 16 > gpurun_out/r4closing/e2e_cap16.json 2>/dev/null; echo "e2e cap16 rc=$?"
for f in e2e e2e_cap16; do python -c "
import json; d=json.load(open('gpurun_out/r4closing/$f.json')); c=d['config']; print('$f', d['value'], d['steps'], c['fetch_ms'], c['pushes_in_flight'], c['events_per_s_while_discovering_keys'], c['events_per_s_all_keys_known'], c['keys_interned'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])"; done
timeout 900 python bench.py > gpurun_out/r4closing/bench_n1.json 2> gpurun_out/r4closing/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4closing/bench_n1.json')); r=d['roofline']; print(d['config']['algo'], d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['stream_read_probe_GBps'], d['tile_major']['frac'], d['tile_major']['traffic'][0], d['secondary']['roofline']['frac'], d['secondary']['roofline']['traffic'], d['secondary']['tile_major']['frac'], d['c5']['value'], d['v2']['roofline']['frac'], d['cpu_baseline']['gpu_matches_cpu_full_log'])"
