#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run37; mkdir -p $OUT
timeout 600 python -m pytest tests/test_upn_gpu.py tests/test_fp8_gpu.py -m gpu -q -s --timeout 400 -k "wrapper_contract or quantize" > $OUT/pytest.log 2>&1; grep -v "^$" $OUT/pytest.log | tail -12 | cut -c1-250
