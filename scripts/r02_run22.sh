#!/bin/bash
# images per packed pass: tile-count quantisation of the LLM down projection (168 tiles at 8 images, 248 at 12)
OUT=$(pwd)/gpurun_out/r02_run22; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 12 16 10; do
timeout 600 python bench.py --batch $b --no-cpu-baseline --steps 12 --warmup 3 > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err; tail -2 $OUT/bench_b$b.err
python - <<P
import json
d=json.load(open('gpurun_out/r02_run22/bench_b$b.json'))
print($b, {k:d[k] for k in ('value','ms_per_step','one_pass_at_a_time')}, d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_gemm_tiles'], d['decode'].get('batched'))
P
done
