#!/bin/bash
# End-of-round evidence, second pass (after the dwconv kernel and the parked GC): the profile set + the GPU suite subset that the last changes touch.
OUT=$(pwd)/gpurun_out/r02_final2; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/r02_profiles.sh r02_final2/prof 2>&1 | tail -32
timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
