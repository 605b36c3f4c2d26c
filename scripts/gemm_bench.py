"""Micro-benchmark of fo1_gemm_bf16 variants on the hot-path shapes (GPU box only).
usage: gemm_bench.py <out.json> [cold|warm] [filter]
`cold` (default) rotates the weight operand through enough distinct copies (> 640 MB) that every launch streams its
weights from HBM, as in the pipeline (8 GB of weights per step never stay in the 256 MB Infinity Cache); `warm` re-uses
one copy (L2 / Infinity-Cache resident), which flatters the small latency-bound GEMMs by ~2x."""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

SHAPES = [  # (name, M, N, K)
    ("llm_qkv", 515, 2560, 2048), ("llm_o", 515, 2048, 2048), ("llm_gateup", 515, 22016, 2048),
    ("llm_down", 515, 2048, 11008), ("vit_qkv", 1564, 3840, 1280), ("vit_proj", 1564, 1280, 1280),
    ("vit_gateup", 1564, 6912, 1280), ("vit_down", 1564, 1280, 3456), ("davit_s2_fc1", 1200, 4096, 1024),
    ("davit_s2_fc2", 1200, 1024, 4096), ("davit_s2_qkv", 1200, 3072, 1024), ("davit_s2_proj", 1200, 1024, 1024),
    ("davit_s3_fc1", 300, 8192, 2048), ("merger1", 391, 5120, 5120), ("fpn3x3_l0", 25024, 512, 4608),
    ("davit_s0_fc1", 19200, 1024, 256), ("sq4096", 4096, 4096, 4096),
]
mode = sys.argv[2] if len(sys.argv) > 2 else "cold"
flt = sys.argv[3] if len(sys.argv) > 3 else ""
VARIANTS = [(st, t, sp) for st in (2, 3, 4, 6) for t in (1, 2, 3) for sp in (1, 2, 4, 8)] + [(2, 4, 1), (0, 0, 0)]
res = []
for name, M, N, K in SHAPES:
    if flt and flt not in name:
        continue
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ncopy = 1 if mode == "warm" else max(2, min(96, int(640e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for staging, tile, splits in VARIANTS:
        if staging >= 2 and K % 64 != 0:
            continue
        if splits > 1 and (M > 2048 or K < 1024):
            continue
        if staging == 6 and tile != 3:
            continue
        if staging in (3, 4) and tile == 4:
            continue
        L.load().fo1_gemm_set_variant(staging, tile)
        L.load().fo1_gemm_set_splitk(splits)
        for i in range(3):
            ops.gemm(a, ws[i % ncopy], out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 24 if M * N * K < 1e11 else 6
        e0.record()
        for i in range(iters):
            ops.gemm(a, ws[(i + 3) % ncopy], out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        res.append(dict(shape=name, M=M, N=N, K=K, staging=staging, tile=tile, splits=splits, us=round(ms * 1e3, 2), tflops=round(tf, 1), mode=mode))
        print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d} staging={staging} tile={tile} splitk={splits}: {ms*1e3:9.2f} us  {tf:8.1f} TF/s", flush=True)
    del ws
L.load().fo1_gemm_set_variant(0, 0)
L.load().fo1_gemm_set_splitk(0)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_bench.json", "w"))
