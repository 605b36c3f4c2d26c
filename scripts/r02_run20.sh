#!/bin/bash
# per-shape GEMM rows of the default workload
OUT=$(pwd)/gpurun_out/r02_run20; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py --profile-shapes --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_shapes.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run20/bench_shapes.json'))
r=d['roofline']
rows=sorted(r['per_step_ms'].items(), key=lambda kv:-kv[1])
for k,v in rows[:70]:
    print(f"{k:60s} {v:8.3f} ms  x{r['launches'].get(k,0)}")
P
