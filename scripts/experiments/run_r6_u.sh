export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/experiments/probe_window.py 2>/dev/null | tee gpurun_out/r06_stream_probe_windows.jsonl
