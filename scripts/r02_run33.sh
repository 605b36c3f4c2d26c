#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run33; mkdir -p $OUT
timeout 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q --timeout 300 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log | cut -c1-220
timeout 300 python scripts/gemm_bench_fp8.py $OUT/gemm_bench_fp8.json 2>&1 | tee $OUT/gemm_bench_fp8.log | cut -c1-260
