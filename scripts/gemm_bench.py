"""Micro-benchmark of fo1_gemm_bf16 variants on the hot-path shapes (GPU box only).
Prints TFLOP/s per (shape, staging, tile); used to pick the dispatch defaults."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

SHAPES = [  # (name, M, N, K)
    ("llm_qkv", 515, 2560, 2048), ("llm_o", 515, 2048, 2048), ("llm_gateup", 515, 22016, 2048),
    ("llm_down", 515, 2048, 11008), ("lm_head", 1, 151936, 2048), ("fpn3x3_l0", 25024, 512, 4608), ("davit_s3_fc1", 300, 8192, 2048), ("vit_qkv", 1564, 3840, 1280), ("vit_proj", 1564, 1280, 1280),
    ("vit_gateup", 1564, 6848, 1280), ("vit_down", 1564, 1280, 3456), ("merger1", 391, 5120, 5120),
    ("davit_s0_fc1", 19200, 1024, 256), ("davit_s2_qkv", 1200, 3072, 1024), ("sq4096", 4096, 4096, 4096),
    ("sq8192", 8192, 8192, 8192),
]
res = []
for name, M, N, K in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for staging, tile, splits in [(2, t, sp) for t in (1, 2, 3) for sp in (1, 4)] + [(3, t, sp) for t in (1, 2, 3) for sp in (1, 4)] + [(0, 0, 0)]:
        if True:
            if staging >= 2 and K % 64 != 0:
                continue
            if splits > 1 and (M > 2048 or K < 1024):
                continue
            L.load().fo1_gemm_set_variant(staging, tile)
            L.load().fo1_gemm_set_splitk(splits)
            for _ in range(3):
                ops.gemm(a, w, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20 if M * N * K < 1e11 else 5
            e0.record()
            for _ in range(iters):
                ops.gemm(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            res.append(dict(shape=name, M=M, N=N, K=K, staging=staging, tile=tile, splits=splits, us=round(ms * 1e3, 2), tflops=round(tf, 1)))
            print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d} staging={staging} tile={tile} splitk={splits}: {ms*1e3:9.2f} us  {tf:8.1f} TF/s", flush=True)
L.load().fo1_gemm_set_variant(0, 0)
L.load().fo1_gemm_set_splitk(0)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_bench.json", "w"))
