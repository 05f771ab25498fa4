# round 4: the whole GPU suite, smoke(), the allocation probe
mkdir -p gpurun_out/r4full
timeout 200 scripts/experiments/vram_probe2 > gpurun_out/r4full/vram_probe2.txt 2>&1; echo "vram_probe2 rc=$?"; cat gpurun_out/r4full/vram_probe2.txt
timeout 1700 python -m pytest tests -q -m gpu -x > gpurun_out/r4full/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r4full/pytest.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4full/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4full/smoke.log
