export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; tail -4 gpurun_out/r06_bench_n1.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value %.4e'%d['value'], 'frac', round(r['frac'],4), 'one_shot', round(r.get('frac_one_shot',0),4))
e=d.get('e2e',{})
print('e2e value %.4e'%e.get('value',0), 'steps', e.get('steps'), 'ms/step', e.get('ms_per_step'), 'parity', e['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'])
c=e.get('config',{})
print('fetches', c.get('fetches'), 'events_timed', c.get('events_timed'), 'host cpu', c.get('host_cpu_ms_per_1e6_records'), c.get('host_cpu_ms_per_1e6_records_by_thread'), 'gen_s', c.get('generate_s'), 'parity_s', c.get('parity_s'))
print({k:(v.get('value') if isinstance(v,dict) else None) for k,v in e.items() if k in ('in_place_8_threads','framing_by_copy_12_threads','mixed_topic')})
PY
