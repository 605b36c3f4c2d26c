"""Mean ALGORITHMIC HBM bytes per launch of the 256 x 256 GEMM over one packed pass, from a `bench.py --profile-shapes` file
(profiles/r05_fused_qkv_and_implicit_conv_per_shape.json), to set beside the PMC traffic per launch of profiles/r05_pmc_traffic.json.
Per launch: A once (an implicit convolution reads its padded map once: M x Cin, not M x 9 Cin) + W once + C once (a SwiGLU epilogue writes N / 2
columns); residual operands of the residual epilogues are NOT counted (so the figure is a lower bound on what any schedule must move).
    python scripts/gemm_algorithmic_bytes.py [per_shape.json] [pmc_traffic.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_fused_qkv_and_implicit_conv_per_shape.json")
pmc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")
d = json.load(open(shapes))["all_on"]
tot = n = flop = 0
for k, ms in d["per_step_ms"].items():
    m = re.match(r"gemm (\d+)x(\d+)x(\d+) t256x256 (\w+)", k)
    if not m:
        continue
    M, N, K = map(int, m.groups()[:3])
    c = d["launches"][k]
    a = M * (K // 9 if m.group(4) == "conv" else K) * 2
    out = M * N * (1 if N in (22016, 6912) else 2)          # gate/up of the LLM / the ViT: SwiGLU in the epilogue, N / 2 bf16 columns written
    tot += (a + N * K * 2 + out) * c
    n += c
    flop += 2 * M * N * K * c
t = json.load(open(pmc))["kernels"]["gemm_bt_p4<256,256>"]["hbm_bytes_per_launch"]
print(json.dumps(dict(launches_per_pass=n, algorithmic_mb_per_launch=round(tot / n / 1e6, 1), gflop_per_launch=round(flop / n / 1e9, 1),
                      pmc_hbm_mb_per_launch=round(t / 1e6, 1), traffic_over_algorithmic=round(t / (tot / n), 3))))
