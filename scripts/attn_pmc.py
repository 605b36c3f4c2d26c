"""Attention shapes of the 25-image pass, one launch each per iteration (GPU box only): the target of the rocprofv3 --pmc passes in
scripts/attn_pmc.sh.  usage: attn_pmc.py [iters]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import ops

B = 25
CASES = [("llm_causal", 651, 16, 2, 128, None, True), ("vit_full", 1564, 16, 16, 80, None, False), ("vit_win", 1564, 16, 16, 80, "win", False)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, L, H, KV, D, seg, causal in CASES:
    T = B * L
    qkv = torch.randn(T, (H + 2 * KV) * D, device="cuda").bfloat16()
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(KV * D, Tp, dtype=torch.bfloat16, device="cuda")
    ops.transpose_into(qkv[:, (H + KV) * D:], vt, 0)
    segs = []
    for b in range(B):
        if seg == "win":
            segs += [(b * L + a, b * L + min(a + 64, L)) for a in range(0, L, 64)]
        else:
            segs.append((b * L, (b + 1) * L))
    items = ops.make_items(segs, "cuda", block=64)
    out = torch.empty(T, H * D, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], vt, items, H, KV, D, 1 / math.sqrt(D), causal, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], vt, items, H, KV, D, 1 / math.sqrt(D), causal, out=out)
    e1.record(); torch.cuda.synchronize()
    fl = 4.0 * H * D * sum((e - s) * (e - s) for s, e in segs) * (0.5 if causal else 1.0)
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:12s} wgs={items.shape[0] * H:6d} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
