mkdir -p gpurun_out/r4j
echo skip tests
t0=$(date +%s); timeout 900 python bench.py > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python - <<'P'
import json
d=json.load(open('gpurun_out/r4j/bench.json'))
r=d['roofline']; print('headline', d['config']['algo'], d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), r.get('stream_read_probe_GBps'))
print('one_shot', d['one_shot'])
print('tile_major', {k:v for k,v in d['tile_major'].items() if k!='one_shot'}, d['tile_major'].get('one_shot'))
s=d['secondary']; print('C2', s['config']['algo'], s['ms_per_step'], s['roofline']['frac'], s['roofline'].get('traffic'), 'tiled', s['tile_major']['frac'], s['tile_major']['traffic'])
print('c5', d['c5'].get('value'), d['c5'].get('ms_per_step'), d['c5'].get('skipped'))
print('v2', d['v2'].get('value'), d['v2'].get('skipped'), (d['v2'].get('roofline') or {}).get('frac'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_matches_cpu_full_log'))
P
