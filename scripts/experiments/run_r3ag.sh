mkdir -p gpurun_out/r3ag
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3ag/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3ag/smoke.log
timeout 600 python bench.py --workload c5 > gpurun_out/r3ag/bench_c5.json 2> gpurun_out/r3ag/bench_c5.err; echo "c5 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3ag/bench_c5.json')); c=d['config']; print('c5', d['value'], c['batch_latency_ms'], c['snapshot_ms'], c['ingest_only_events_per_sec_synced_per_batch'], d['cpu_baseline'])"
