#!/bin/bash
# usage: scripts/prof.sh <tag>   (run on the GPU box via gpurun; writes gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
find $OUT -type f | head -40; du -sh $OUT; tail -5 $OUT/trace.log
python $REPO/scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
