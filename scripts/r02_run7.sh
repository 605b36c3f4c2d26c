#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run7; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/decode_batch_profile.py 8 > $OUT/decode_b8.log 2>&1; cat $OUT/decode_b8.log | grep -v amdgpu.ids
timeout 600 python scripts/decode_batch_profile.py 1 > $OUT/decode_b1.log 2>&1; cat $OUT/decode_b1.log | grep -v amdgpu.ids
