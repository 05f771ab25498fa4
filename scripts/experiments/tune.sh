for le in 8 16; do for g in 1 2 4 8; do echo "rows LE=$le G=$g: $(SURGE_REPLAY_LE_ROWS=$le SURGE_REPLAY_ROWS_GROUPS=$g python scripts/experiments/variants.py 2>&1 | grep 'algo=3')"; done; done
for kb in 64 128 256 512 1024; do echo "TASK_KB=$kb: $(SURGE_REPLAY_TASK_KB=$kb python scripts/experiments/variants.py 2>&1 | grep -E 'algo=(1|2)' | tr '\n' ' ')"; done
