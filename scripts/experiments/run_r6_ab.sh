export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ingest_gpu.py tests/test_store.py -x -q -m gpu --timeout 200 2>&1 | tail -4
SURGE_INGEST_CHAIN=lane timeout 1200 python -m pytest tests/test_ingest_gpu.py -x -q -m gpu --timeout 200 -k "chain or fixed16 or play_json" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_bench_rehearsal.py -x -q -m gpu -k "e2e" --timeout 300 2>&1 | tail -3
bash scripts/experiments/run_r6_z.sh 2>&1 | tee gpurun_out/r06_section_chain_ab.txt
