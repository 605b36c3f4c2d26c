#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run36; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "persistent or gemm" --timeout 200 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
timeout 400 python scripts/gemm_t0_study.py $OUT/gemm_t0_study.json 2>&1 | tee $OUT/gemm_t0_study.log | python -c "
import sys,ast
for l in sys.stdin:
    if l.startswith('{'):
        d=ast.literal_eval(l); print(d['shape'], d['rounds'], 'normal',d['normal_us'],d['normal_again_us'],'persistent',d['persistent_us'],d['persistent_again_us'],'nostore',d['no_stores_us'],'1k',d['one_k_tile_us'],d['one_k_tile_no_stores_us'])
"
