#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_9; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_batched_decode_gpu.py tests/test_stage_abi_gpu.py tests/test_decode_pool_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -8 > $OUT/pytest_subset.log
FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab.json 16 17 25 32 > $OUT/decode_ab.log 2>&1
tail -4 $OUT/pytest_subset.log; grep "^==" $OUT/decode_ab.log; grep "attn_decode" $OUT/decode_ab.log
