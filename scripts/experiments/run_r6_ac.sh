export TMPDIR=/tmp
export SURGE_REPLAY_LIB=$PWD/surge_amd/libsurge_replay_exp.so
timeout 600 python bench.py --workload e2e --steps 10 --warmup 2 --parity none --serial-framing 2>&1 >/dev/null | grep -i "experiments\|error" | head
