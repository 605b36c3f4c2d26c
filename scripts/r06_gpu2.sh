#!/bin/bash
# round 6, GPU call 2: Infinity-Cache prefetch experiment + HBM counter calibration
export HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_2; mkdir -p $OUT
timeout 900 python scripts/r06_mall_prefetch.py $OUT/mall_prefetch.json > $OUT/mall_prefetch.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o c -- python $ROOT/scripts/pmc_calibrate.py run > $OUT/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o c -- python $ROOT/scripts/pmc_calibrate.py run > $OUT/cal_write.log 2>&1
cd $ROOT
python scripts/pmc_calibrate.py fold $OUT/cal_fetch $OUT/cal_write $OUT/pmc_calibration.json > $OUT/pmc_calibration.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/mall_prefetch.log | tail -8; cat $OUT/pmc_calibration.log
