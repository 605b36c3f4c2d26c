#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run6; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_abi.py -m gpu -q --timeout 800 > $OUT/pytest_dec.log 2>&1; tail -30 $OUT/pytest_dec.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print("value", d["value"], "ms/step", d["ms_per_step"]); print("decode", d["decode"]); print("single", d["one_image_at_a_time"], d.get("one_pass_at_a_time"))
r=d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm_tiles"]); print(r.get("hfre"))
print(d["stage_kernel_ms"])
PY
