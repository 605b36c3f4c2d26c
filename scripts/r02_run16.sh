#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run16; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_llm_gpu.py tests/test_dropin_gpu.py -q --timeout 600 > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
for cfg in "8 0" "8 1" "4 0" "1 0"; do set -- $cfg
FO1_GEMV_RPL=$2 timeout 600 python scripts/decode_batch_profile.py $1 > $OUT/decode_b$1_rpl$2.log 2>&1; grep -v amdgpu $OUT/decode_b$1_rpl$2.log | head -13
done
