#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run26; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_upn_gpu.py tests/test_msda_gpu.py -m gpu -q --timeout 600 -s > $OUT/pytest.log 2>&1; grep -v amdgpu $OUT/pytest.log | tail -30
