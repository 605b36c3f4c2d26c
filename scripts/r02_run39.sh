#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run39; mkdir -p $OUT
timeout 300 python scripts/sustained.py 150 25 2 nogc 2>&1 | grep -v amdgpu.ids | tee $OUT/sustained_b25_nogc.log | grep -v smi | cut -c1-200
