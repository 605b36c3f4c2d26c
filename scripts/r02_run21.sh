#!/bin/bash
# fast epilogue activations: full suite + per-shape bench
OUT=$(pwd)/gpurun_out/r02_run21; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_all.log 2>&1; tail -12 $OUT/pytest_all.log
timeout 900 python bench.py --profile-shapes --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_shapes.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run21/bench_shapes.json'))
print({k:d[k] for k in ('value','ms_per_step','one_image_at_a_time','one_pass_at_a_time')})
r=d['roofline']
print(r['kernel'],r['achieved'],r['frac'],r['all_gemm_tiles'])
rows=sorted(r['per_step_ms'].items(), key=lambda kv:-kv[1])
for k,v in rows[:34]:
    print(f"{k:60s} {v:8.3f} ms  x{r['launches'].get(k,0)}")
P
