mkdir -p gpurun_out/r3ab
timeout 600 python scripts/ingest_gpu_bench.py 8000 > gpurun_out/r3ab/ingest_gpu.json 2> gpurun_out/r3ab/ingest_gpu.err; echo "ingest rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3ab/ingest_gpu.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'host', round(v['host_decoder']['records_per_sec']/1e6,1), 'framing+dev', round(v['framing_plus_device_decoder']['records_per_sec']/1e6,1), 'framing_s', round(v['framing_plus_device_decoder']['host_framing_s'],3), 'push_s', round(v['framing_plus_device_decoder']['device_push_s'],3), v['framing_plus_device_decoder']['equal_to_host_decoder'])
"
timeout 600 python bench.py --workload e2e > gpurun_out/r3ab/bench_e2e.json 2> gpurun_out/r3ab/bench_e2e.err; echo "e2e rc=$?"; tail -c 2500 gpurun_out/r3ab/bench_e2e.json; tail -3 gpurun_out/r3ab/bench_e2e.err
timeout 900 python scripts/auto_policy_sweep.py > gpurun_out/r3ab/auto_sweep.jsonl 2> gpurun_out/r3ab/auto_sweep.err; echo "sweep rc=$?"; cat gpurun_out/r3ab/auto_sweep.jsonl; tail -3 gpurun_out/r3ab/auto_sweep.err
