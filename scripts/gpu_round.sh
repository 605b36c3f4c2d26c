#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/.
# usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tee $OUT/pytest_gpu.log | tail -40
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tee $OUT/smoke.log | tail -5
echo "== bench"; timeout 600 python bench.py 2>&1 | tee $OUT/bench.log | tail -3
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/rocprof.log 2>&1
cd $ROOT
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f; done
# keep the merged-back payload small: drop the raw per-dispatch trace if it is huge
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
