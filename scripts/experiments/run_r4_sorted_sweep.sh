mkdir -p gpurun_out/r4sweep
for cfg in "7 16" "8 16" "7 16" "8 16" "7 16" "8 16"; do
set -- $cfg
SURGE_REPLAY_SORTED_WAVES=$1 SURGE_REPLAY_LE_SORTED=$2 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --parity none > gpurun_out/r4sweep/w$1_le$2.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r4sweep/w$1_le$2.json')); r=d['roofline']; print('waves/CU $1 LE $2:', d['config']['algo'], r['kernel_ms_min_median_max'], round(r['frac'],4), r.get('stream_read_probe_GBps'))"
done
