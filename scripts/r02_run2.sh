#!/bin/bash
# round 2, GPU call 2: correctness + race screen of the 256x256 ping-pong GEMM, then its micro-benchmark on the batched shapes
OUT=$(pwd)/gpurun_out/r02_run2; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_ops_gpu.py -k "p8" -m gpu -q --timeout 500 > $OUT/pytest_p8.log 2>&1; tail -15 $OUT/pytest_p8.log
timeout 900 python scripts/gemm_bench_p8.py $OUT/gemm_bench_p8.json 8 > $OUT/gemm_bench_p8.log 2>&1; tail -95 $OUT/gemm_bench_p8.log
