mkdir -p gpurun_out
timeout 1500 python scripts/prof_traffic.py r02_c3_10Magg_sorted > gpurun_out/prof_c3.log 2>&1; tail -4 gpurun_out/prof_c3.log
timeout 900 python scripts/prof_traffic.py r02_c2_rows --workload c2 > gpurun_out/prof_c2.log 2>&1; tail -4 gpurun_out/prof_c2.log
timeout 900 python scripts/prof_traffic.py r02_c4shard_1250k_chunked --aggregates 1250000 > gpurun_out/prof_c4s.log 2>&1; tail -4 gpurun_out/prof_c4s.log
timeout 900 python scripts/prof_traffic.py r02_zipf_300k_chunked --aggregates 300000 > gpurun_out/prof_300k.log 2>&1; tail -3 gpurun_out/prof_300k.log
timeout 900 python scripts/prof_traffic.py r02_zipf_100k_flat --aggregates 100000 > gpurun_out/prof_100k.log 2>&1; tail -3 gpurun_out/prof_100k.log
ls gpurun_out/prof_r02_*
