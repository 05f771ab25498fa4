# round 4 closing run: the default bench line, PMC traffic of the four fold configurations the line quotes, kernel-level
# evidence for bytes -> states, the e2e variants
mkdir -p gpurun_out/r4final
t0=$(date +%s); timeout 900 python bench.py > gpurun_out/r4final/bench_n1.json 2> gpurun_out/r4final/bench_n1.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
PROF_PASS_TIMEOUT=300 timeout 1500 python scripts/prof_traffic.py r04_c3_10Magg_sorted > gpurun_out/r4final/prof_c3_sorted.log 2>&1; echo "prof c3 sorted rc=$?"
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=300 timeout 1500 python scripts/prof_traffic.py r04_c3_10Magg_tiled --algo tiled > gpurun_out/r4final/prof_c3_tiled.log 2>&1; echo "prof c3 tiled rc=$?"
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=200 timeout 900 python scripts/prof_traffic.py r04_c2_rows --workload c2 > gpurun_out/r4final/prof_c2_rows.log 2>&1; echo "prof c2 rows rc=$?"
PROF_SKIP_SQ=1 PROF_PASS_TIMEOUT=200 timeout 900 python scripts/prof_traffic.py r04_c2_tiled --workload c2 --algo tiled > gpurun_out/r4final/prof_c2_tiled.log 2>&1; echo "prof c2 tiled rc=$?"
PROF_PASSES=trace,sq1,sq2 PROF_PASS_TIMEOUT=600 timeout 1800 python scripts/prof_ingest.py r04_e2e_c3pop --workload e2e > gpurun_out/r4final/prof_e2e.log 2>&1; echo "prof e2e rc=$?"
PROF_PASSES=trace PROF_PASS_TIMEOUT=600 timeout 900 python scripts/prof_ingest.py r04_e2e_c3pop_serial --workload e2e --serial-framing --events-cap 2 > gpurun_out/r4final/prof_e2e_serial.log 2>&1; echo "prof e2e serial rc=$?"
timeout 600 python bench.py --workload e2e --no-capacity-hint > gpurun_out/r4final/e2e_no_hint.json 2>/dev/null; echo "e2e no hint rc=$?"
timeout 600 python bench.py --workload e2e --codec none > gpurun_out/r4final/e2e_uncompressed.json 2>/dev/null; echo "e2e none rc=$?"
tail -3 gpurun_out/prof_r04_c3_10Magg_sorted/*summary.txt gpurun_out/prof_r04_c3_10Magg_tiled/*summary.txt gpurun_out/prof_r04_c2_rows/*summary.txt gpurun_out/prof_r04_c2_tiled/*summary.txt 2>/dev/null | grep -i "traffic per launch"
