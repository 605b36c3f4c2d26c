#!/bin/bash
# GPU-box visit for op-level parity + GEMM micro-benchmark.  usage: bash scripts/gpu_ops.sh [tag]
TAG=${1:-ops}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu (ops)"; timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 300 2>&1 | tee $OUT/pytest_ops.log | tail -60
echo "== gemm bench"; timeout 600 python scripts/gemm_bench.py $OUT/gemm_bench.json 2>&1 | tee $OUT/gemm_bench.log | tail -80
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
