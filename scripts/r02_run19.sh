#!/bin/bash
# hardware bf16 conversion everywhere: full suite, decode A/B, default bench line
OUT=$(pwd)/gpurun_out/r02_run19; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_all.log 2>&1; tail -12 $OUT/pytest_all.log
timeout 900 python scripts/decode_ab.py $OUT/decode_ab.json 1 8 16 > $OUT/decode_ab.log 2>&1; grep -v amdgpu $OUT/decode_ab.log | grep -E "^==|identical|Error|error|Traceback|gemv_" | head -60
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r02_run19/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','one_image_at_a_time','one_pass_at_a_time')})
print(d['decode'])
r=d['roofline']; print(r['kernel'],r['achieved'],r['frac'],r['all_gemm_tiles']); print(r['per_step_ms']); print(r['hfre'])
print(d['stage_kernel_ms'])
P
