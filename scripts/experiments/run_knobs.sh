for cfg in "1250000 1792" "1250000 2048" "1250000 2304" "1250000 2560" "800000 990" "800000 1536" "800000 2048" "2000000 2048" "2000000 2470" "2000000 3072" "500000 616" "500000 1024" "500000 2048"; do
set -- $cfg
echo "== n=$1 CHUNK_T=$2"; ZIPF_AGGS=$1 FOLDS=30 ALGO=5 SURGE_REPLAY_CHUNK_T=$2 timeout 120 python scripts/experiments/variance_probe.py 2>&1 | grep "^rep 1" | cut -c1-120
done
