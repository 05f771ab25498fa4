export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "inplace2:--framing-threads 2" "copy12:--framing-by-copy --framing-threads 12"; do
  tag=${v%%:*}; flags=${v#*:}
  SURGE_DBG_TIMING=1 timeout 600 python bench.py --workload e2e --steps 6 --warmup 2 --parity none $flags > /dev/null 2> gpurun_out/r06_e2e_timing_$tag.err
  echo "== $tag"; grep "stage1" gpurun_out/r06_e2e_timing_$tag.err | tail -40 | awk '{a[$4]+=$5; n[$4]++} END {for (k in a) printf "%s %.1f us (n=%d)\n", k, a[k]/n[k], n[k]}'
done
