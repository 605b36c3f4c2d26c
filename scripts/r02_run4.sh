#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run4; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
cp gpurun_out/fulldepth_metrics.json $OUT/ 2>/dev/null
for cfg in "8 1" "8 2"; do
  set -- $cfg
  timeout 600 python bench.py --batch $1 --inflight $2 --no-cpu-baseline > $OUT/bench_b$1_i$2.json 2> $OUT/bench_b$1_i$2.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_b$1_i$2.json").read())
    r=d["roofline"]
    print("batch $1 inflight $2: value", round(d["value"],1), "img/s  ms/step", round(d["ms_per_step"],2), "one_pass", d.get("one_pass_at_a_time"), "dom", r["kernel"], r["achieved"], "allgemm", r["all_gemm_tiles"])
    print("   stage", d["stage_kernel_ms"])
except Exception as e:
    print("batch $1 inflight $2 FAILED", e); print(open("$OUT/bench_b$1_i$2.err").read()[-1500:])
PY
done
