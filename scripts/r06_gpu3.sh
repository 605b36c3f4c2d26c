#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_3; mkdir -p $OUT
timeout 900 python scripts/r06_two_pools.py $OUT/two_pools.json > $OUT/two_pools.log 2>&1
tail -6 $OUT/two_pools.log
