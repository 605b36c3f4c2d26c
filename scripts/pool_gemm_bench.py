#!/usr/bin/env python3
"""Decode-pool projections at P = 64 / 128 rows, cold weights (GPU box only): every tile / staging / split-K form of fo1_gemm_bf16 at M = P
(round 4's first run also timed a dedicated weights-to-VGPR kernel: profiles/r04_pool_gemm_stream_kernel_vs_tile_kernels.json).  usage: pool_gemm_bench.py <out.json>"""
import os
os.environ.setdefault("FO1_AB", "1")
import json
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

SHAPES = [("qkv", 2560, 2048, "qkv"), ("o", 2048, 2048, "res"), ("gateup", 22016, 2048, "swiglu"), ("down", 2048, 11008, "res"), ("lm_head", 151936, 2048, "plain")]
res = []
lib = L.load()
for P in (128, 64):
    for name, N, K, kind in SHAPES:
        x = (torch.randn(P, K, device="cuda") * 0.5).bfloat16()
        ncopy = max(2, min(48, int(700e6 / (N * K * 2)) + 1))
        ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
        r = (torch.randn(P, N, device="cuda")).bfloat16()
        nw = torch.ones(N, device="cuda").bfloat16()
        st = torch.zeros(P, 8, dtype=torch.int32)
        st[:, 0] = torch.arange(P) * 8
        st = st.cuda()
        cos = torch.ones(1024, 128, device="cuda").bfloat16()
        kc = torch.zeros(2, 2048, 128, dtype=torch.bfloat16, device="cuda")
        vt = torch.zeros(256, 2048, dtype=torch.bfloat16, device="cuda")
        wb = N * K * 2

        def timeit(fn, iters=20):
            for i in range(3):
                fn(ws[i % ncopy])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn(ws[(i + 3) % ncopy])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        def tile(w):
            if kind == "swiglu":
                return ops.gemm(x, w, act=ops.ACT_SWIGLU16)
            if kind == "res":
                return ops.gemm(x, w, residual=r)
            return ops.gemm(x, w)

        variants = [(0, 0, 0)] + [(stg, t, sp) for stg in (3, 4) for t in (1, 2, 3) for sp in ((1,) if kind in ("swiglu", "plain") else (1, 2, 4, 8, 16))]
        for stg, t, sp in variants:
            if N * K > 2e8 and (stg, t) not in ((0, 0), (3, 1), (4, 1), (3, 3)):
                continue
            lib.fo1_gemm_set_variant(stg, t)
            lib.fo1_gemm_set_splitk(sp)
            try:
                us = timeit(tile, iters=12)
            except Exception as e:
                print("skip", name, stg, t, sp, str(e)[:80])
                continue
            res.append(dict(P=P, shape=name, impl=f"tile staging={stg} tile={t} splitk={sp}", us=round(us, 2), tbps=round(wb / us / 1e6, 3)))
            print(res[-1], flush=True)
        lib.fo1_gemm_set_variant(0, 0)
        lib.fo1_gemm_set_splitk(0)
        del ws
        torch.cuda.empty_cache()
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pool_gemm_bench.json", "w"), indent=0)
