export TMPDIR=/tmp
export SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_exp.so
for mode in "0 lane" "0 wave" "2 lane" "2 wave"; do
set -- $mode
echo "== dbg $1 chain $2"
SURGE_SECTION_TICKS=1 SURGE_DBG_DECODE=$1 SURGE_INGEST_CHAIN=$2 timeout 600 python bench.py --workload e2e --steps 6 --warmup 2 --parity none --serial-framing 2>&1 >/dev/null | grep "experiments" | tail -4
done
