mkdir -p gpurun_out/r3ac
timeout 600 python bench.py --workload e2e > gpurun_out/r3ac/bench_e2e.json 2> gpurun_out/r3ac/bench_e2e.err; echo "e2e rc=$?"; tail -c 2600 gpurun_out/r3ac/bench_e2e.json; tail -3 gpurun_out/r3ac/bench_e2e.err
timeout 600 python scripts/ingest_gpu_bench.py 8000 > gpurun_out/r3ac/ingest_gpu.json 2> gpurun_out/r3ac/ingest_gpu.err; echo "ingest rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3ac/ingest_gpu.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'host', round(v['host_decoder']['records_per_sec']/1e6,1), 'framing+dev', round(v['framing_plus_device_decoder']['records_per_sec']/1e6,1), 'framing_s', round(v['framing_plus_device_decoder']['host_framing_s'],3), 'push_s', round(v['framing_plus_device_decoder']['device_push_s'],3), v['framing_plus_device_decoder']['equal_to_host_decoder'])
"
cd /tmp && export TMPDIR=/tmp
PROF_PASS_TIMEOUT=200 timeout 1300 python $GRAFT_REPO_ROOT/scripts/prof_traffic.py r03_c3_10Magg_tiled > $GRAFT_REPO_ROOT/gpurun_out/r3ac/prof_c3.log 2>&1; echo "prof c3 rc=$?"; tail -4 $GRAFT_REPO_ROOT/gpurun_out/r3ac/prof_c3.log
PROF_PASS_TIMEOUT=120 PROF_SKIP_SQ=1 timeout 700 python $GRAFT_REPO_ROOT/scripts/prof_traffic.py r03_c4shard_1250k_tiled --aggregates 1250000 --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/r3ac/prof_c4s.log 2>&1; echo "prof c4s rc=$?"; tail -4 $GRAFT_REPO_ROOT/gpurun_out/r3ac/prof_c4s.log
