#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_6; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_batched_decode_gpu.py tests/test_decode_pool_gpu.py tests/test_stage_abi_gpu.py tests/test_ops_gpu.py tests/test_llm_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -8 > $OUT/pytest_subset.log
FO1_DECODE_CHUNKS="64:2048" timeout 600 python scripts/r06_decode_ab.py $OUT/decode_ab.json 1 25 > $OUT/decode_ab.log 2>&1
FO1_AB=1 timeout 600 python scripts/pool_bench.py --slots 128 64 > $OUT/pool_bench.json 2> $OUT/pool_bench.err
tail -4 $OUT/pytest_subset.log; grep "^==" $OUT/decode_ab.log; grep "attn_decode" $OUT/decode_ab.log; python -c "
import json; d=json.load(open('$OUT/pool_bench.json')); print({k:(v['ms_per_step'], v.get('kernels_ms_per_step',{}).get('attn_decode_one_chunk')) for k,v in d.items() if isinstance(v,dict) and 'ms_per_step' in v})"
