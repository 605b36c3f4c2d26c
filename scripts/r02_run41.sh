#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run41; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_towers_gpu.py tests/test_golden_towers_gpu.py tests/test_fulldepth_parity_gpu.py -m gpu -q -x --timeout 400 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --main-only --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.err | cut -c1-200
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run41/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_gemm_tiles'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:6]}, d['stage_kernel_ms'])
P
