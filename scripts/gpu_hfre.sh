#!/bin/bash
TAG=${1:-hfre}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hfre_gpu.py -m gpu -q --timeout 300 2>&1 | tail -5
for n in 32 100; do python scripts/hfre_only.py $n 50 2>&1 | tail -1 | tee -a $OUT/hfre_time.log; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o hfre -- python $ROOT/scripts/hfre_only.py 32 50 > $OUT/ktrace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o hfre -- python $ROOT/scripts/hfre_only.py 32 5 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o hfre -- python $ROOT/scripts/hfre_only.py 32 5 > $OUT/pmc_write.log 2>&1
cd $ROOT
grep -h "hfre" $(find $OUT/ktrace -name "*kernel_stats.csv") | cut -c1-160
for d in pmc_fetch pmc_write; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); echo $d $f; head -1 $f; grep hfre_pool $f | head -3; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
