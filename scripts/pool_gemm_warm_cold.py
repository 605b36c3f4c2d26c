#!/usr/bin/env python3
"""Decode-pool projections at M = 128: the product dispatch with COLD weights (a rotation of copies larger than the 256 MB Infinity Cache, what a
decode step sees) against WARM weights (one copy, resident in the Infinity Cache / L2) — what a weight prefetch ahead of the step could buy.
usage: pool_gemm_warm_cold.py [out.json]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import ops

SHAPES = [("qkv", 2560, 2048, "bias"), ("o", 2048, 2048, "res"), ("gateup", 22016, 2048, "swiglu"), ("down", 2048, 11008, "res")]
res = []
P = 128
for name, N, K, kind in SHAPES:
    x = (torch.randn(P, K, device="cuda") * 0.5).bfloat16()
    ncopy = max(2, int(900e6 / (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    r = torch.randn(P, N, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()

    def call(w):
        if kind == "swiglu":
            return ops.gemm(x, w, act=ops.ACT_SWIGLU16)
        if kind == "res":
            return ops.gemm(x, w, residual=r)
        return ops.gemm(x, w, bias)

    def timeit(pick, iters=40):
        for i in range(4):
            call(pick(i))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            call(pick(i + 4))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    cold = timeit(lambda i: ws[i % ncopy])
    warm = timeit(lambda i: ws[0])
    row = dict(shape=name, N=N, K=K, weight_mb=round(N * K * 2 / 1e6, 1), cold_us=round(cold, 2), warm_us=round(warm, 2), cold_tbps=round(N * K * 2 / cold / 1e6, 2), warm_tbps=round(N * K * 2 / warm / 1e6, 2))
    res.append(row)
    print(json.dumps(row), flush=True)
    del ws
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
