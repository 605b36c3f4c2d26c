#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run25; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/gemm_bench_p8.py $OUT/gemm_persist_ab.json 12 2>&1 | grep -v amdgpu | tee $OUT/gemm_persist_ab.log
