#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run30; mkdir -p $OUT
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_llm_gpu.py tests/test_stage_abi_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python scripts/decode_ab.py $OUT/decode_ab.json 1 8 > $OUT/decode_ab.log 2>&1; grep -E "^==|gemv_mfma|identical" $OUT/decode_ab.log | cut -c1-200
