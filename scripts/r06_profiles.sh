#!/bin/bash
# Round-6 evidence on ONE box: the default bench line (compact contract line on stdout, full record via --json-out), rocprofv3 kernel stats of the same
# command's packed passes — graph replay with two passes in flight, graph replay one pass at a time, and EAGER one pass at a time (bench.py's own
# per-kernel figures come from an eager profiled pass: the like-for-like row of the agreement table; VERDICT r5 #2c) —, PMC passes (FETCH_SIZE /
# WRITE_SIZE / MFMA busy, separately), the per-shape GEMM record the algorithmic read / write bytes come from, the decode-pool step per kernel.
TAG=${1:-r06_profiles}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python bench.py --json-out $OUT/bench_full.json > $OUT/bench.json 2> $OUT/bench.err; tail -c 1400 $OUT/bench.json
FO1_AB=1 timeout 600 python bench.py --main-only --no-cpu-baseline --steps 6 --warmup 2 --dataset none --profile-shapes --json-out $OUT/bench_per_shape.json > /dev/null 2> $OUT/per_shape.err
FO1_AB=1 timeout 600 python scripts/pool_bench.py --slots 64 128 > $OUT/pool_step.json 2> $OUT/pool_step.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --steps 12 --dataset none --json-out "" > $OUT/rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --inflight 1 --steps 12 --dataset none --json-out "" > $OUT/rocprof1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1e -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --inflight 1 --eager --steps 6 --warmup 2 --dataset none --json-out "" > $OUT/rocprof1e.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only --dataset none --json-out "" > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only --dataset none --json-out "" > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only --dataset none --json-out "" > $OUT/pmc_mfma.log 2>&1
cd $ROOT
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/prof1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_inflight1.csv
cp $(find $OUT/prof1e -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_inflight1_eager.csv
python scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json $OUT/bench_per_shape.json
python scripts/mfma_busy_summary.py $OUT/pmc_mfma $OUT/mfma_busy.json
echo "== bench (eager profiled pass) vs rocprofv3, EAGER one pass at a time" > $OUT/agreement.txt
python scripts/check_profile_agreement.py $OUT/bench_full.json $OUT/kernel_stats_inflight1_eager.csv >> $OUT/agreement.txt
echo "== bench (eager profiled pass) vs rocprofv3, graph replay one pass at a time" >> $OUT/agreement.txt
python scripts/check_profile_agreement.py $OUT/bench_full.json $OUT/kernel_stats_inflight1.csv >> $OUT/agreement.txt
cat $OUT/agreement.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
