#!/usr/bin/env python3
"""Where the 256 x 256 GEMM's main loop spends its time (GPU box, A/B library): the product kernel against instantiations with the LDS-DMA
issue (1), the fragment reads (2), the barriers (4) or the vmcnt waits (8) of the K loop removed, or every DMA piece read from K tile 0 (16:
L2-hot sources), on the packed pass's big products.
Ablated launches compute garbage; only their durations are used.  usage: gemm_loop_ablation.py [out.json]"""
import os
os.environ.setdefault("FO1_AB", "1")
import json
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

lib = L.load()
SHAPES = [("vit_qkv", 39100, 3840, 1280), ("vit_fc", 39100, 6912, 1280), ("llm_o", 16275, 2048, 2048), ("llm_gateup_plain", 16275, 22016, 2048), ("llm_down", 16275, 2048, 11008), ("square", 8192, 8192, 8192)]
ABLS = (0, 64, 1, 8, 16, 64, 0)     # 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no vmcnt waits, 16 DMA from K tile 0 only, 24 = 8 + 16, 32 = half of each wave's DMA pieces issued in its load-Y segment instead of its MFMA segments (valid results); 64 = only the A half of the DMA pieces issued; 0 and 64 are timed first AND last
res = []
for name, M, N, K in SHAPES:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(3)]
    row = dict(shape=name, M=M, N=N, K=K)
    for abl in ABLS:
        lib.fo1_gemm_set_debug(abl << 6)
        for i in range(3):
            ops.gemm(x, ws[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            ops.gemm(x, ws[i % 3])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        key = f"abl{abl}" if f"abl{abl}_us" not in row else f"abl{abl}_again"
        row[f"{key}_us"] = round(us, 1)
        row[f"{key}_tflops"] = round(2.0 * M * N * K / us / 1e6, 1)
    lib.fo1_gemm_set_debug(0)
    tiles = -(-M // 256) * -(-N // 256)
    rounds = -(-tiles // 256)
    row["us_per_k_tile_round"] = {f"abl{a}": round(row[f"abl{a}_us"] / rounds / (K // 64), 3) for a in sorted(set(ABLS))}
    res.append(row)
    print(json.dumps(row), flush=True)
    del x, ws
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
