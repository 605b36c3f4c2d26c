#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_4; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_batched_decode_gpu.py tests/test_decode_pool_gpu.py tests/test_stage_abi_gpu.py tests/test_e2e_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -8 > $OUT/pytest_subset.log
FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab.json 1 2 25 > $OUT/decode_ab.log 2>&1
FO1_AB=1 timeout 600 python scripts/pool_bench.py --slots 128 > $OUT/pool_bench.json 2> $OUT/pool_bench.err
for P in 1 2 3; do
  timeout 600 python bench.py --steps 10 --warmup 2 --e2e-passes 24 --e2e-only --decode-pools $P --no-cpu-baseline > $OUT/e2e_pools$P.json 2> $OUT/e2e_pools$P.err
done
tail -4 $OUT/pytest_subset.log; grep "^==" $OUT/decode_ab.log; for P in 1 2 3; do tail -c 900 $OUT/e2e_pools$P.json; echo; done
