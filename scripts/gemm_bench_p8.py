"""Micro-benchmark of the large-M GEMM paths on the batched-prefill shapes (GPU box only): the 256x256 ping-pong kernel
(tile 5) against the best of the round-1 tiles (auto), cold weights (every launch streams its weight panel from HBM).
usage: gemm_bench_p8.py <out.json> [images]"""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S, Lp = 1564, 651
SHAPES = [
    ("vit_qkv", B * S, 3840, 1280), ("vit_proj", B * S, 1280, 1280), ("vit_gateup", B * S, 6912, 1280), ("vit_down", B * S, 1280, 3456),
    ("llm_qkv", B * Lp, 2560, 2048), ("llm_o", B * Lp, 2048, 2048), ("llm_gateup", B * Lp, 22016, 2048), ("llm_down", B * Lp, 2048, 11008),
    ("davit_s0_fc1", B * 19200, 1024, 256), ("davit_s1_fc1", B * 4800, 2048, 512), ("davit_s2_qkv", B * 1200, 3072, 1024),
    ("davit_s2_fc1", B * 1200, 4096, 1024), ("davit_s2_fc2", B * 1200, 1024, 4096), ("davit_s3_fc1", B * 300, 8192, 2048),
    ("fpn3x3_l0", B * 25024, 512, 4608), ("merger1", B * 391, 5120, 5120),
    ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
]
VARIANTS = [("p4_persistent", 0, 5, 1, 1 | 4), ("p4_one_tile", 0, 5, 1, 1), ("p4_persistent", 0, 5, 1, 1 | 4), ("p4_one_tile", 0, 5, 1, 1)]
res = []
for name, M, N, K in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ncopy = max(2, min(64, int(640e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for label, staging, tile, splits, sched in VARIANTS:
        L.load().fo1_gemm_set_big_schedule(sched)
        L.load().fo1_gemm_set_variant(staging, tile)
        L.load().fo1_gemm_set_splitk(splits)
        for i in range(3):
            ops.gemm(a, ws[i % ncopy], out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20 if M * N * K < 3e11 else 6
        e0.record()
        for i in range(iters):
            ops.gemm(a, ws[(i + 3) % ncopy], out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        res.append(dict(shape=name, M=M, N=N, K=K, variant=label, us=round(ms * 1e3, 2), tflops=round(tf, 1)))
        print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d} {label:13s}: {ms*1e3:9.2f} us  {tf:8.1f} TF/s", flush=True)
    del ws
L.load().fo1_gemm_set_variant(0, 0)
L.load().fo1_gemm_set_splitk(0)
L.load().fo1_gemm_set_big_schedule(1)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_bench_p8.json", "w"))
