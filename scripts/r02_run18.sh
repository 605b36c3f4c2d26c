#!/bin/bash
# decode v2, second pass: branch-free MFMA GEMV pipeline (counted vmcnt), split attention default again
OUT=$(pwd)/gpurun_out/r02_run18; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_stage_abi_gpu.py -m gpu -q --timeout 600 > $OUT/pytest_decode.log 2>&1; tail -15 $OUT/pytest_decode.log
timeout 900 python scripts/decode_ab.py $OUT/decode_ab.json 1 8 16 > $OUT/decode_ab.log 2>&1; grep -v amdgpu $OUT/decode_ab.log | grep -E "^==|identical|Error|error|Traceback|gemv_" | head -60
