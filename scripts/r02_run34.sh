#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run34; mkdir -p $OUT
timeout 600 python -m pytest tests/test_fp8_engine_gpu.py -m gpu -q -s --timeout 500 > $OUT/pytest.log 2>&1; tail -45 $OUT/pytest.log | cut -c1-200
timeout 600 python bench.py --fp8 mlp --no-cpu-baseline --main-only --steps 12 --warmup 3 > $OUT/bench_fp8.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run34/bench_fp8.json'))
print({k:d[k] for k in ('value','ms_per_step','dtype')}, d['roofline']['kernel'], d['roofline']['achieved'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:8]})
P
