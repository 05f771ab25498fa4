export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err ) 2>&1 | tail -4
tail -3 gpurun_out/r06_bench_n1.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_n1.json').read().strip().splitlines()[-1])
print('value %.4e'%d['value'], 'ms/step', d['ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('frac','kernel','kernel_ms','frac_one_shot','frac_one_shot_wall','stream_read_probe_GBps','traffic')})
print('one_shot', {k:d['one_shot'][k] for k in ('index_build_ms','first_fold_kernel_ms','prepare_wall_ms')})
print('cpu_baseline', d['cpu_baseline'])
print('tile_major', d['tile_major'].get('frac'), 'secondary', d['secondary'].get('roofline',{}).get('frac'), d['secondary'].get('tile_major',{}).get('frac') if isinstance(d['secondary'].get('tile_major'),dict) else None)
print('c5', d['c5'].get('value'), d['c5'].get('config',{}).get('batch_latency_ms'))
print('v2', d['v2'].get('roofline',{}).get('frac'))
c4=d['c4_shard']; print('c4_shard', c4.get('roofline',{}).get('frac'), c4.get('roofline',{}).get('frac_one_shot_wall'), c4.get('one_shot'), c4.get('cpu_baseline',{}).get('gpu_matches_cpu_full_shard'))
e=d['e2e']
if 'skipped' in e: print('e2e skipped', e)
else:
    print('e2e %.3e'%e['value'], 'steps', e['steps'], 'parity', e['cpu_baseline']['gpu_states_match_cpu_fold_of_the_source_events'], {k:e['config'].get(k) for k in ('host_cpu_ms_per_1e6_records','framing_cpu_ms_per_1e6_records','receive_copy_cpu_ms_per_1e6_records','framing_threads','framing')})
    print('layouts', e.get('layouts_events_per_s'))
    print('bound_log', e.get('bound_log'))
    print('by_copy', e.get('framing_by_copy_12_threads'))
PY
