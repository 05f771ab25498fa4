mkdir -p gpurun_out/r3ao
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4
timeout 400 python scripts/flat_small_sweep.py > gpurun_out/r3ao/flat_sweep.jsonl 2> gpurun_out/r3ao/flat_sweep.err; echo "sweep rc=$?"
python - <<'P'
import json
for line in open('gpurun_out/r3ao/flat_sweep.jsonl'):
    r=json.loads(line); print(r['shape'], r['aggregates'])
    for k,v in r.items():
        if isinstance(v,dict): print('   ', k, v)
P
PROF_PASS_TIMEOUT=120 timeout 700 python scripts/prof_traffic.py r03_zipf_100k_flat --aggregates 100000 --algo flat 2>&1 | grep -v amdgpu.ids | tail -3
for le in 16 8; do
  SURGE_REPLAY_LE_FLAT=$le timeout 300 python bench.py --workload c5 > gpurun_out/r3ao/c5_le$le.json 2> gpurun_out/r3ao/c5_le$le.err; echo "c5 le=$le rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r3ao/c5_le$le.json')); print('c5 le=$le', d['value'], json.dumps(d['config'])[:600])"
done
