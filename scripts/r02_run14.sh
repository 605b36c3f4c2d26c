#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run14; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_llm_gpu.py tests/test_dropin_gpu.py -q --timeout 600 > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python scripts/decode_batch_profile.py 8 > $OUT/decode_b8.log 2>&1; grep -v amdgpu $OUT/decode_b8.log
timeout 600 python scripts/decode_batch_profile.py 1 > $OUT/decode_b1.log 2>&1; grep -v amdgpu $OUT/decode_b1.log
