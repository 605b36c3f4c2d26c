#!/usr/bin/env python3
"""Round 6: the decode pool's weight-streaming GEMMs at P = 128 / 64 rows with the deep-ring tiles (gemm.hip launch_ring_deep), cold weights
(a rotation of weight copies larger than the Infinity Cache).  GPU box only.  usage: r06_pool_gemm.py <out.json>"""
import os
os.environ.setdefault("FO1_AB", "1")
import json
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

SHAPES = [("gateup", 22016, 2048, "swiglu"), ("lm_head", 151936, 2048, "plain"), ("down", 2048, 11008, "planes"), ("qkv", 2560, 2048, "planes"), ("o", 2048, 2048, "planes")]
res = []
lib = L.load()
for P in (128, 64):
    for name, N, K, kind in SHAPES:
        x = (torch.randn(P, K, device="cuda") * 0.5).bfloat16()
        ncopy = max(2, min(48, int(700e6 / (N * K * 2)) + 1))
        ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
        part = torch.empty(16 * P * N, dtype=torch.float32, device="cuda") if kind == "planes" else None
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

        if kind == "planes":
            splits_list = (4, 8, 16) if name == "down" else (2, 4)
            variants = [(0, 0, sp) for sp in splits_list] + [(stg, 7, sp) for stg in (3, 4, 6) for sp in splits_list]
        else:
            variants = [(0, 0, 0), (3, 1, 0), (3, 7, 0), (4, 7, 0), (6, 7, 0), (3, 6, 0), (4, 6, 0), (5, 6, 0)]
        ref = None
        for stg, t, sp in variants:
            if P == 64 and t in (6, 7):
                continue
            lib.fo1_gemm_set_variant(stg, t)
            try:
                if kind == "planes":
                    def run(w):
                        return ops.gemm_partials(x, w, sp, part)
                    s_eff = run(ws[0])
                    torch.cuda.synchronize()
                    out = part[:s_eff * P * N].view(s_eff, P, N).sum(0)
                else:
                    def run(w):
                        return ops.gemm(x, w, act=ops.ACT_SWIGLU16) if kind == "swiglu" else ops.gemm(x, w)
                    out = run(ws[0]).float()
                torch.cuda.synchronize()
                if ref is None or (kind == "planes" and (stg, t) == (0, 0)):
                    ref = out.clone()
                    err, same = 0.0, True
                else:
                    err = float((out - ref).abs().max() / ref.abs().max())
                    same = bool(torch.equal(out, ref))
                us = timeit(run, iters=12 if N * K > 2e8 else 24)
            except Exception as e:
                print("skip", name, stg, t, sp, str(e)[:120], flush=True)
                continue
            res.append(dict(P=P, shape=name, staging=stg, tile=t, splits=sp, us=round(us, 2), tbps=round(wb / us / 1e6, 3), rel_err_vs_first=err, bitwise_equal=same))
            print(res[-1], flush=True)
        lib.fo1_gemm_set_variant(0, 0)
        del ws
        torch.cuda.empty_cache()
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_pool_gemm.json", "w"), indent=0)
