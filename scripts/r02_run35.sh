#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run35; mkdir -p $OUT
for B in 24 25 20; do
timeout 400 python bench.py --batch $B --no-cpu-baseline --main-only --steps 8 --warmup 3 > $OUT/bench_b$B.json 2> $OUT/bench_b$B.err; tail -1 $OUT/bench_b$B.err | cut -c1-200
python - <<P
import json
d=json.load(open('gpurun_out/r02_run35/bench_b$B.json'))
print($B, {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_gemm_tiles'])
P
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
