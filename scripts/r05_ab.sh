#!/bin/bash
# round 5 A/B on the GPU box: fused q/k/v epilogue on / off, GEMM tile-order group size (main-only bench lines, 20 steps each)
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --main-only --steps 20 --no-cpu-baseline > gpurun_out/r05_ab_$tag.json 2> gpurun_out/r05_ab_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_ab_$tag.json")); r=d["roofline"]["per_step_ms"]
    print("$tag", "value", round(d["value"],2), "ms", round(d["ms_per_step"],2), "gemm", r.get("gemm_bt_p4<256,256>"), "qkv_post", r.get("qkv_post_vit"), r.get("qkv_post_llm"), "rms", r.get("rmsnorm"))
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r05_ab_$tag.err").read()[-1500:])
PY
}
run fused_on  X=1
run fused_off FO1_QKV_FUSED=0
run ab_gm8 FO1_AB=1
run ab_gm4 FO1_AB=1 FO1_GEMM_GROUP_M=4
run ab_gm2 FO1_AB=1 FO1_GEMM_GROUP_M=2
