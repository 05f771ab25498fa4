mkdir -p gpurun_out
PROF_PASS_TIMEOUT=200 timeout 1200 python scripts/prof_traffic.py r03_c3_10Magg_tiled > gpurun_out/prof_c3.log 2>&1; tail -5 gpurun_out/prof_c3.log
PROF_PASS_TIMEOUT=120 timeout 700 python scripts/prof_traffic.py r03_c4shard_1250k_tiled --aggregates 1250000 --steps 100 > gpurun_out/prof_c4s.log 2>&1; tail -5 gpurun_out/prof_c4s.log
ls gpurun_out/prof_r03_*
