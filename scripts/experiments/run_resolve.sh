mkdir -p gpurun_out/resolve
timeout 600 python -m pytest tests/test_ingest_gpu.py tests/test_store.py tests/test_event_decode.py tests/test_frame_gpu.py -q -m gpu > gpurun_out/resolve/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/resolve/pytest.log
grep -n "passed\|failed\|FAILED\|Error\|assert\|pytest rc" gpurun_out/resolve/pytest.log | tail -8
timeout 600 python bench.py --workload e2e > gpurun_out/resolve/bench_e2e.json 2> gpurun_out/resolve/bench_e2e.err; echo "e2e rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/resolve/bench_e2e.json')); print('e2e', d['value'], d['config']['fetch_ms'], d['config']['host_framing_ms_per_fetch'], d['config']['device_decode_groupby_fold_ms_per_fetch'], d['cpu_baseline']['gpu_states_match_cpu_fold_of_the_decoded_events'])"
timeout 300 python scripts/ingest_gpu_bench.py 4000000 > gpurun_out/resolve/ingest_gpu.json 2> gpurun_out/resolve/ingest_gpu.err; echo "ingest rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/resolve/ingest_gpu.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'host_decoder' in v:
        print(k, 'host', round(v['host_decoder']['records_per_sec']/1e6,1))
        for l in v:
            if l.startswith('framing'): print('   ', l, round(v[l]['records_per_sec']/1e6,1), 'framing_s', round(v[l]['host_framing_s'],3), 'push_s', round(v[l]['device_push_s'],3), v[l]['equal_to_host_decoder'])
"
timeout 400 python scripts/ingest_gpu_mt_bench.py 6 > gpurun_out/resolve/ingest_mt.json 2> gpurun_out/resolve/ingest_mt.err; echo "mt rc=$?"; grep -v amdgpu gpurun_out/resolve/ingest_mt.err | tail -6; tail -c 1500 gpurun_out/resolve/ingest_mt.json
