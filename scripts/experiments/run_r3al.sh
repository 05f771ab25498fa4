mkdir -p gpurun_out/r3al
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3al/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3al/pytest.log
grep -v amdgpu.ids gpurun_out/r3al/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3al/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3al/smoke.log
timeout 600 python bench.py > gpurun_out/r3al/bench_n1.json 2> gpurun_out/r3al/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3al/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source','')[:60], d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_matches_cpu_full_log'), d.get('one_shot'))"
PROF_PASS_TIMEOUT=150 timeout 900 python scripts/prof_traffic.py r03_v2_ledger_2Magg --workload v2 2>&1 | grep -v amdgpu.ids | tail -8; echo "prof v2 rc=$?"
