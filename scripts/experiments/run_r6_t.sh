export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in libsurge_replay.so libsurge_replay_exp2.so; do
SURGE_REPLAY_LIB=$PWD/surge_amd/$lib timeout 600 python scripts/experiments/tiled_transport_probe.py 2>/dev/null
done | tee gpurun_out/r06_tiled_transport_probe.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['lib'][:24], d['algo'], 'waves', d['waves_per_cu'], 'subs', d['subs'], 'ms %.3f'%d['ms'], 'frac %.4f'%d['frac'])
"
