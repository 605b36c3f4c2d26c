"""Micro-benchmark of fo1_attention_bf16 for the hot-path shapes and query-block sizes (GPU box only)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import ops
CASES = [("llm_causal_L515", 515, 16, 2, 128, None, True), ("llm_causal_L850", 850, 16, 2, 128, None, True),
         ("vit_full_S1564", 1564, 16, 16, 80, None, False), ("vit_win_S1564", 1564, 16, 16, 80, "win", False),
         ("davit_s0_win", 144 * 154, 8, 8, 32, "w144", False), ("davit_s2_win", 144 * 12, 32, 32, 32, "w144", False)]
for name, L, H, KV, D, seg, causal in CASES:
    qkv = torch.randn(L, (H + 2 * KV) * D, device="cuda").bfloat16()
    Lp = (L + 63) // 64 * 64
    vt = torch.zeros(KV * D, Lp, dtype=torch.bfloat16, device="cuda")
    ops.transpose_into(qkv[:, (H + KV) * D:], vt, 0)
    if seg == "win":
        segs = [(a, min(a + 64, L)) for a in range(0, L, 64)]
    elif seg == "w144":
        segs = [(a, a + 144) for a in range(0, L, 144)]
    else:
        segs = [(0, L)]
    for blk in (64, 32, 16):
        items = ops.make_items(segs, "cuda", block=blk)
        out = torch.empty(L, H * D, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], vt, items, H, KV, D, 1 / math.sqrt(D), causal, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], vt, items, H, KV, D, 1 / math.sqrt(D), causal, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:18s} q_block={blk:2d} wgs={items.shape[0]*H:6d}: {e0.elapsed_time(e1)/20*1e3:8.1f} us", flush=True)
