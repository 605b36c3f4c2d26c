#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run29; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hfre_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 300 python scripts/hfre_ab.py $OUT/hfre_ab.json 2>&1 | tee $OUT/hfre_ab.log | grep finish_vec | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --main-only --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run29/bench.json'))
print(d['value'], d['roofline']['hfre'])
P
