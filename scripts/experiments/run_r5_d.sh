#!/bin/bash
# Round 5, fourth run: micro-batch buffers of the engine grow with room to spare; one against two consumer threads; the chunked
# kernel without scratch on the C4 shard and on the whole C3 log beside SORTED; the GPU suite.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5d; mkdir -p $O
t0=$(date +%s)
lap() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $O/rc.log; }
for K in 512 64 0; do
  SURGE_BENCH_TRACE=1 timeout 300 python bench.py --workload e2e --steps 28 --warmup 2 --txn-flush-events $K > $O/e2e_k$K.json 2> $O/e2e_k$K.err; lap "e2e K=$K rc=$?"
done
SURGE_BENCH_TRACE=1 timeout 300 python bench.py --workload e2e --steps 28 --warmup 2 --txn-flush-events 512 --two-thread-consumer > $O/e2e_k512_2t.json 2> $O/e2e_k512_2t.err; lap "e2e K=512 two threads rc=$?"
python - <<'P'
import json
O="gpurun_out/r5d"
for n in ("k512","k64","k0","k512_2t"):
    try:
        d=json.loads([l for l in open(f"{O}/e2e_{n}.json") if l.startswith("{")][-1]); c=d["config"]
        print(n, "%.4g ev/s"%d["value"], "ms/step %.2f"%d["ms_per_step"], "parity", d["cpu_baseline"]["gpu_states_match_cpu_fold_of_the_source_events"], "fetch_ms", c["fetch_ms"], "framing ms", round(c["host_framing_ms_per_fetch"],2), "finish+fold", round(c["finish_and_fold_ms_per_fetch"],2), "push", round(c["push_async_host_ms_per_fetch"],2), "disc %.3g known %.3g"%(c["events_per_s_while_discovering_keys"] or 0, c["events_per_s_all_keys_known"] or 0))
    except Exception as e: print(n, "failed", e)
P
grep "\[bench\] fetch" $O/e2e_k512.err | awk '{print $3, $5, $9}' | tr '\n' ';'; echo
timeout 300 python bench.py --workload c4-shard --no-cpu-baseline > $O/c4shard.json 2> $O/c4shard.err; lap "c4-shard rc=$?"
timeout 300 python bench.py --workload c4-shard --no-cpu-baseline --algo sorted > $O/c4shard_sorted.json 2> $O/c4shard_sorted.err; lap "c4-shard sorted rc=$?"
timeout 300 python bench.py --no-secondary --no-cpu-baseline --algo chunked > $O/c3_chunked.json 2> $O/c3_chunked.err; lap "c3 chunked rc=$?"
timeout 300 python bench.py --no-secondary --no-cpu-baseline --algo sorted > $O/c3_sorted.json 2> $O/c3_sorted.err; lap "c3 sorted rc=$?"
python - <<'P'
import json
O="gpurun_out/r5d"
for n in ("c4shard","c4shard_sorted","c3_chunked","c3_sorted"):
    try:
        d=json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1]); r=d["roofline"]
        print(n, d["config"]["algo"], "frac %.4f"%r["frac"], "kernel_ms", r["kernel_ms_min_median_max"])
    except Exception as e: print(n, "failed", e)
P
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; lap "pytest rc=$?"
tail -n 4 $O/pytest.log
