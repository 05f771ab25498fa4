# the full config C4 (10 M aggregates, 8 ranks) as a FUNCTIONAL run on one GPU: every rank on cuda:0, stub transport
mkdir -p gpurun_out/final
python -c "
import sys; sys.path.insert(0,'tests')
from test_comm import build_rccl_stub; print(build_rccl_stub())"
export SURGE_BENCH_REHEARSAL=1 SURGE_RCCL_LIBRARY=$PWD/tests/rccl_stub/librccl_stub.so SURGE_RCCL_STUB_DIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/final/rehearsal_c4_8ranks.json 2> gpurun_out/final/rehearsal_c4_8ranks.err
echo "rc=$?"; tail -3 gpurun_out/final/rehearsal_c4_8ranks.err; cut -c1-1500 gpurun_out/final/rehearsal_c4_8ranks.json
