#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run12; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hfre_gpu.py -q --timeout 600 > $OUT/pytest_hfre.log 2>&1; tail -5 $OUT/pytest_hfre.log
timeout 900 python scripts/hfre_sweep.py > $OUT/sweep.log 2>&1; grep -v amdgpu $OUT/sweep.log | tail -60
cp gpurun_out/hfre_sweep.json $OUT/ 2>/dev/null
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_batched_prefill_gpu.py tests/test_dropin_gpu.py -q --timeout 600 > $OUT/pytest_e2e.log 2>&1; tail -5 $OUT/pytest_e2e.log
