#!/bin/bash
# persistent 256x256 GEMM: bit-identity vs one tile per workgroup, GEMM unit tests, then A/B of the bench
OUT=$(pwd)/gpurun_out/r02_run24; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 500 -k "gemm" > $OUT/pytest_gemm.log 2>&1; tail -12 $OUT/pytest_gemm.log
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > $OUT/bench_persist.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run24/bench_persist.json'))
print({k:d[k] for k in ('value','ms_per_step','one_pass_at_a_time')}, d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['all_gemm_tiles'])
print({k:v for k,v in sorted(d['roofline']['per_step_ms'].items(), key=lambda kv:-kv[1])[:8]})
P
