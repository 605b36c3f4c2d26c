#!/bin/bash
OUT=$(pwd)/gpurun_out/r02_run10; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print("value", d["value"], "ms/step", d["ms_per_step"]); print("decode", d["decode"]); print("single", d["one_image_at_a_time"], d.get("one_pass_at_a_time"))
r=d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["all_gemm_tiles"])
print(d["stage_kernel_ms"])
for k,v in r["per_step_ms"].items(): print(f"  {k:28s} {v:8.3f} x{r['launches'][k]}")
PY
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print(json.dumps(d["roofline"].get("hfre"), indent=1))
PY
FO1_HFRE_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --batch 1 --inflight 1 --steps 20 > $OUT/bench_b1_3launch.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --batch 1 --inflight 1 --steps 20 > $OUT/bench_b1_fused.json 2>/dev/null
python - <<PY
import json
for n in ("bench_b1_3launch","bench_b1_fused"):
    d=json.loads(open("$OUT/%s.json"%n).read()); h=d["roofline"].get("hfre"); print(n, d["value"], h and {k:h[k] for k in ("kernel","us_all_launches","frac","achieved")})
PY
