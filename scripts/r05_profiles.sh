#!/bin/bash
# Round-5 evidence on ONE box: the default bench line (value, end_to_end through the decode pool, driver_level + driver_level_countbench, hires incl.
# its end-to-end leg, decode.pool, cpu leg), rocprofv3 kernel stats of the same command's packed passes (two in flight, and one pass at a time whose
# per-kernel durations are not stretched by a concurrent pass), PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately), MFMA-busy pass.
TAG=${1:-r05_profiles}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --steps 12 > $OUT/rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --inflight 1 --steps 12 > $OUT/rocprof1.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_mfma.log 2>&1
cd $ROOT
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/prof1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_inflight1.csv
python scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
python scripts/mfma_busy_summary.py $OUT/pmc_mfma $OUT/mfma_busy.json
python scripts/check_profile_agreement.py $OUT/bench.json $OUT/kernel_stats_inflight1.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
