#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_10; mkdir -p $OUT
for T in 1 2 3 4; do
FO1_ATTN_TILES=$T FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab_t$T.json 17 25 32 > $OUT/decode_ab_t$T.log 2>&1
echo "tiles $T"; grep "^==" $OUT/decode_ab_t$T.log; grep "attn_decode_split" $OUT/decode_ab_t$T.log
done
