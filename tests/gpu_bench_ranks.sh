#!/bin/bash
# Development aid: bench.py as N rank processes on this box's one GPU through the mock librccl
# (tests/mock_rccl), at the full c2 size by default:   bash tests/gpu_bench_ranks.sh [N] [ng] [steps]
N=${1:-6}; NG=${2:-64}; STEPS=${3:-5}
R=$(cd "$(dirname "$0")/.." && pwd)
export GHIP_RCCL_LIB=$R/tests/mock_rccl/librccl_mock.so BENCH_TRANSPORT=rccl MASTER_ADDR=127.0.0.1
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port 29517 "$R/bench.py" --gpus $N --steps $STEPS --warmup 2 --ng $NG
