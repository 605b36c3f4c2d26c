"""A/B of the two prefill attention kernels on the bench pass's shapes (round 5): attn_fwd_kernel (16x16 MFMA, 64 queries per
workgroup) against attn_fwd32_kernel (32x32 MFMA, 8 waves x 32 queries).  Prints one JSON line per shape.

    python scripts/attn32_ab.py [--images 25] [--iters 20]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlm_fo1_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=25)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(0)
    B = a.images
    shapes = [
        dict(name="llm_prefill_651x%d" % B, H=16, KV=2, D=128, seg=652, causal=True, blks=(64, 128, 256)),
        dict(name="vit_full_1564x%d" % B, H=16, KV=16, D=80, seg=1564, causal=False, blks=(64, 256)),
        dict(name="hires_llm_3072x4", H=16, KV=2, D=128, seg=3072, causal=True, blks=(64, 128), n=4),
        dict(name="hires_vit_9216x4", H=16, KV=16, D=80, seg=9216, causal=False, blks=(64, 256), n=4),
    ]
    for sh in shapes:
        n = sh.get("n", B)
        H, KV, D, S = sh["H"], sh["KV"], sh["D"], sh["seg"]
        L = n * S
        segs = [(i * S, (i + 1) * S) for i in range(n)]
        qkv = (torch.randn(L, (H + 2 * KV) * D, device="cuda") * 1.0).to(torch.bfloat16)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
        vt = torch.zeros(KV * D, (L + 63) // 64 * 64, dtype=torch.bfloat16, device="cuda")
        ops.transpose_into(v, vt, 0)
        out = torch.empty(L, H * D, dtype=torch.bfloat16, device="cuda")
        flops = 4.0 * H * D * n * (S * (S + 1) / 2.0 if sh["causal"] else float(S) * S)
        row = dict(shape=sh["name"], gflop=round(flops / 1e9, 1))
        ref = None
        for blk in sh["blks"]:
            if blk == 128 and (H // KV) % 2:
                continue
            items = ops.make_items(segs, "cuda", block=blk)
            us = timed(lambda: ops.attention(q, k, vt, items, H, KV, D, 1.0 / math.sqrt(D), sh["causal"], out=out), a.iters)
            row["q%d_us" % blk] = round(us, 1)
            row["q%d_tflops" % blk] = round(flops / us / 1e6, 1)
            o = out.float()
            if ref is None:
                ref = o.clone()
            else:
                row["q%d_max_abs_diff_vs_q64" % blk] = round(float((o - ref).abs().max()), 5)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
