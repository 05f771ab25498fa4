mkdir -p gpurun_out/bench
timeout 1500 python bench.py > gpurun_out/bench/n1.json 2> gpurun_out/bench/n1.err; echo "rc=$?"; tail -3 gpurun_out/bench/n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench/n1.json'))
print(json.dumps({k:(v if k not in ('cpu_baseline',) else v) for k,v in d.items()}, indent=1)[:6000])
PY
