# round 4, first GPU call: (1) the one-off costs of a first bind on a fresh box, (2) the read-stream probes side by side,
# (3) kernel-level evidence for bytes -> states on the round-3 sources
mkdir -p gpurun_out/r4a
timeout 300 scripts/experiments/vram_probe > gpurun_out/r4a/vram_probe.txt 2>&1; echo "vram_probe rc=$?"; cat gpurun_out/r4a/vram_probe.txt
timeout 200 scripts/experiments/probe2 > gpurun_out/r4a/probe2.txt 2>&1; echo "probe2 rc=$?"; cat gpurun_out/r4a/probe2.txt
PROF_PASSES=trace,sq1,sq2 timeout 900 python scripts/prof_ingest.py r04_e2e_r3sources --workload e2e --steps 6 > gpurun_out/r4a/prof.log 2>&1; echo "prof rc=$?"
tail -60 gpurun_out/prof_r04_e2e_r3sources/r04_e2e_r3sources_summary.txt
