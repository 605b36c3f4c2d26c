#!/bin/bash
# round 6, GPU call 1: sanity of the hidden-visibility build + decode baselines + pool GEMM tiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_product_lib_gpu.py tests/test_ops_gpu.py tests/test_decode_pool_gpu.py tests/test_batched_decode_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5 > $OUT/pytest_subset.log
timeout 600 python scripts/r06_pool_gemm.py $OUT/pool_gemm.json > $OUT/pool_gemm.log 2>&1
FO1_DECODE_CHUNKS="64:2048 256:2048 1024:1024" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab.json 1 25 > $OUT/decode_ab.log 2>&1
FO1_AB=1 timeout 600 python scripts/pool_bench.py --slots 128 > $OUT/pool_bench.json 2> $OUT/pool_bench.err
tail -3 $OUT/pytest_subset.log; tail -5 $OUT/pool_gemm.log; grep "^==" $OUT/decode_ab.log
