#!/bin/bash
# round 5 A/B (3): determinism probe, per-shape rows with the fused q/k/v epilogue and the implicit convolution on / off, tile-order group size
mkdir -p gpurun_out
timeout 200 python scripts/replica_determinism.py 32 > gpurun_out/r05_determinism.json 2> gpurun_out/r05_determinism.err; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r05_determinism.json"))
    for k,v in d["rows"].items(): print(k, v)
except Exception as e:
    print("determinism FAILED", e); print(open("gpurun_out/r05_determinism.err").read()[-1500:])
PY
shape() { tag=$1; shift; env FO1_AB=1 "$@" timeout 300 python bench.py --main-only --profile-shapes --steps 12 --no-cpu-baseline > gpurun_out/r05_shape_$tag.json 2> gpurun_out/r05_shape_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_shape_$tag.json")); r=d["roofline"]; ps=r["per_step_ms"]; ln=r["launches"]
    print("$tag value", round(d["value"],2), "gemm total", round(sum(v for k,v in ps.items() if k.startswith("gemm ")),2))
    for k in sorted(ps, key=lambda k:-ps[k]):
        if ("2560x2048" in k or "3840x1280" in k or "4096x1280" in k or k.startswith("qkv_post") or "conv" in k or "x4608" in k or "x2304" in k or "x9216" in k or k in ("im2col","layernorm")): print("   ", k, ps[k], "x", ln[k], "=", round(ps[k]/ln[k]*1e3,1), "us")
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r05_shape_$tag.err").read()[-1500:])
PY
}
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --main-only --steps 30 --no-cpu-baseline > gpurun_out/r05_ab3_$tag.json 2> gpurun_out/r05_ab3_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_ab3_$tag.json")); r=d["roofline"]["per_step_ms"]
    print("$tag", "value", round(d["value"],2), "ms", round(d["ms_per_step"],2), "gemm", r.get("gemm_bt_p4<256,256>"), "im2col", r.get("im2col"))
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r05_ab3_$tag.err").read()[-1500:])
PY
}
shape all_on X=1
shape all_off FO1_QKV_FUSED=0 FO1_CONV_IMPLICIT=0
run gm8 FO1_AB=1
run gm4 FO1_AB=1 FO1_GEMM_GROUP_M=4
run gm2 FO1_AB=1 FO1_GEMM_GROUP_M=2
