#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run28; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hfre_gpu.py tests/test_stage_abi_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python scripts/hfre_ab.py $OUT/hfre_ab.json 2>&1 | tee $OUT/hfre_ab.log | cut -c1-400
