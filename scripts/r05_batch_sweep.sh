#!/bin/bash
# round 5: images per packed pass x passes in flight around the default (25, 2) after the round's kernel changes (main-only bench lines, 16 steps each)
mkdir -p gpurun_out
for cfg in "25 2" "23 2" "24 2" "26 2" "28 2" "30 2" "32 2" "25 3" "32 3" "16 3"; do
  set -- $cfg
  timeout 200 python bench.py --main-only --no-cpu-baseline --no-hires --dataset none --steps 16 --warmup 4 --batch $1 --inflight $2 > gpurun_out/r05_sweep_b$1_i$2.json 2> gpurun_out/r05_sweep_b$1_i$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_sweep_b$1_i$2.json"))
    print("batch $1 inflight $2: value", round(d["value"],2), "ms/pass", round(d["ms_per_step"],2), "gemm frac", d["roofline"]["frac"])
except Exception as e:
    print("batch $1 inflight $2 FAILED", e); print(open("gpurun_out/r05_sweep_b$1_i$2.err").read()[-800:])
PY
done
