#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run31; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_all.log 2>&1; tail -8 $OUT/pytest_all.log
timeout 600 python bench.py --no-cpu-baseline --main-only --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run31/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['achieved'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:6]})
P
