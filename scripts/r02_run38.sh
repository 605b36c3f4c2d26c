#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run38; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_fp8_engine_gpu.py tests/test_llm_gpu.py tests/test_towers_gpu.py -m gpu -q --timeout 400 > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-250
timeout 600 python bench.py --fp8 all --no-cpu-baseline --main-only --steps 8 --warmup 3 > $OUT/bench_fp8.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run38/bench_fp8.json'))
print({k:d[k] for k in ('value','ms_per_step','dtype')}, d['roofline']['kernel'], d['roofline']['achieved'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:8]})
P
