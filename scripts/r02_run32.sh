#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run32; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_llm_gpu.py tests/test_towers_gpu.py tests/test_batched_decode_gpu.py tests/test_upn_gpu.py tests/test_fulldepth_parity_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline --main-only --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run32/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['achieved'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:6]})
P
