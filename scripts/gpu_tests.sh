#!/bin/bash
# usage: bash scripts/gpu_tests.sh <tag> <pytest args...>   (logs -> gpurun_out/<tag>/)
TAG=$1; shift
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest "$@" -m gpu -q --timeout 600 2>&1 | tee $OUT/pytest.log | tail -70
