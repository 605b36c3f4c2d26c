#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run5; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
cp gpurun_out/fulldepth_metrics.json $OUT/ 2>/dev/null
timeout 900 python scripts/gemm_bench_p8.py $OUT/gemm_bench_p8.json 8 > $OUT/gemm_bench_p8.log 2>&1; tail -60 $OUT/gemm_bench_p8.log
