#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_5; mkdir -p $OUT
timeout 600 python scripts/r06_hfre_order_ab.py $OUT/hfre_order_ab.json > $OUT/hfre_order_ab.log 2>&1
for O in 1 0 1; do
  FO1_AB=1 FO1_HFRE_ORDER=$O timeout 600 python bench.py --main-only --no-cpu-baseline --steps 8 --warmup 2 --dataset none --json-out $OUT/bench_order${O}_full.json > $OUT/bench_order$O.json 2> $OUT/bench_order$O.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_order${O}_full.json"))
print("order $O value", round(d["value"],2), json.dumps(d["roofline"]["hfre"])[:600])
PY
done
timeout 600 python -m pytest tests/test_hfre_gpu.py -m gpu -q -x 2>&1 | tail -3
grep -v "^{'B'" $OUT/hfre_order_ab.log | tail; grep "'B'" $OUT/hfre_order_ab.log | cut -c1-200
