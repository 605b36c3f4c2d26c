#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_11; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_batched_decode_gpu.py tests/test_stage_abi_gpu.py tests/test_decode_pool_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -8 > $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
for T in 1 2 3 4; do
FO1_ATTN_TILES=$T FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab_t$T.json 17 25 32 > $OUT/decode_ab_t$T.log 2>&1
echo "tiles $T"; grep "^==" $OUT/decode_ab_t$T.log; grep "attn_decode_split" $OUT/decode_ab_t$T.log
done
FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab_small.json 1 8 16 > $OUT/decode_ab_small.log 2>&1
grep "^==" $OUT/decode_ab_small.log; grep "attn_decode_split" $OUT/decode_ab_small.log
FO1_AB=1 timeout 600 python scripts/pool_bench.py --slots 128 > $OUT/pool_bench.json 2> $OUT/pool_bench.err
python -c "
import json; d=json.load(open('$OUT/pool_bench.json')); print({k:(v['ms_per_step'], v.get('kernels_ms_per_step',{}).get('attn_decode_one_chunk')) for k,v in d.items() if isinstance(v,dict) and 'ms_per_step' in v})"
