mkdir -p gpurun_out/r4n
PROF_PASSES=sq1,sq2 PROF_PASS_TIMEOUT=600 timeout 1500 python scripts/prof_ingest.py r04_e2e_serial_pmc --workload e2e --events-cap 2 --serial-framing > gpurun_out/r4n/prof.log 2>&1; echo "prof rc=$?"
grep -A1 "^lz4_exec_kernel\|^lz4_parse_kernel\|^section_kernel" gpurun_out/prof_r04_e2e_serial_pmc/r04_e2e_serial_pmc_summary.txt | head -20
