#!/bin/bash
# Round-3 evidence: default bench line, rocprofv3 kernel stats of the same command (and of the one-pass-at-a-time variant, whose
# per-kernel durations are not stretched by a concurrent pass), PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately), MFMA-busy pass.
TAG=${1:-r03_profiles}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --steps 12 > $OUT/rocprof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o bench -- python $ROOT/bench.py --no-cpu-baseline --main-only --inflight 1 --steps 12 > $OUT/rocprof1.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o b -- python $ROOT/bench.py --eager --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --main-only > $OUT/pmc_mfma.log 2>&1
cd $ROOT
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/prof1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_inflight1.csv
python scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
python scripts/mfma_busy_summary.py $OUT/pmc_mfma $OUT/mfma_busy.json
python scripts/check_profile_agreement.py $OUT/bench.json $OUT/kernel_stats_inflight1.csv
tail -3 $OUT/pmc_mfma.log
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
# round-3 extras on the same box: configs[4] (1344^2 x 300 proposals as 3 prompts x 100) in bf16 and fp8, and a second CountBench-like dataset sample
timeout 600 python bench.py --image 1344x1344 --boxes 300 --no-cpu-baseline --main-only > $OUT/bench_hires_bf16.json 2> $OUT/bench_hires_bf16.err
timeout 600 python bench.py --image 1344x1344 --boxes 300 --fp8 --no-cpu-baseline --main-only > $OUT/bench_hires_fp8.json 2> $OUT/bench_hires_fp8.err
cut -c1-200 $OUT/bench_hires_bf16.json; cut -c1-200 $OUT/bench_hires_fp8.json
