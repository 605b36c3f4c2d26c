#!/bin/bash
# perf visit: gemm micro-bench, bench (graph + eager + shape profile), rocprofv3 kernel stats.  usage: gpu_perf.sh <tag>
TAG=${1:-perf}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== gemm bench"; timeout 900 python scripts/gemm_bench.py $OUT/gemm_bench.json > $OUT/gemm_bench.log 2>&1; tail -3 $OUT/gemm_bench.log
echo "== bench graph"; timeout 900 python bench.py --profile-shapes 2>&1 | tail -1 | tee $OUT/bench_graph.json | cut -c1-400
echo "== bench eager"; timeout 900 python bench.py --eager --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_eager.json | cut -c1-300
echo "== rocprofv3"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --eager > $OUT/rocprof.log 2>&1
cd $ROOT
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -40 $f; cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace*.csv" -size +10M -delete
find $OUT/prof -name "*.db" -delete
