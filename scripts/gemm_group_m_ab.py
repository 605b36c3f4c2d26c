"""A/B of the 256 x 256 GEMM's tile order (round 5): tile rows per group of the XCD-grouped order (fo1_gemm_set_group_m) on the
25-image pass's products, cold weights.  An XCD's ~32 concurrent tiles are `rows` tile rows x 32 / rows tile columns.
usage: gemm_group_m_ab.py <out.json> [images]"""
import os
os.environ.setdefault("FO1_AB", "1")
import json, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
S, Lp = 1564, 651
SHAPES = [
    ("vit_qkv", B * S, 3840, 1280, 0), ("vit_proj", B * S, 1280, 1280, 0), ("vit_gateup", B * S, 6912, 1280, 3), ("vit_down", B * S, 1280, 3456, 0),
    ("llm_qkv", B * Lp, 2560, 2048, 0), ("llm_o", B * Lp, 2048, 2048, 0), ("llm_gateup", B * Lp, 22016, 2048, 3), ("llm_down", B * Lp, 2048, 11008, 0),
    ("davit_s2_fc1", B * 1200, 4096, 1024, 1), ("davit_s2_fc2", B * 1200, 1024, 4096, 0),
]
GMS = [8, 2, 4, 16, 32, 8]
res = []
for name, M, N, K, act in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ncopy = max(2, min(64, int(640e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    ref = None
    row = dict(shape=name, M=M, N=N, K=K)
    for gm in GMS:
        L.load().fo1_gemm_set_group_m(gm)
        for i in range(3):
            out = ops.gemm(a, ws[i % ncopy], act=act)
        torch.cuda.synchronize()
        if ref is None:
            ref = ops.gemm(a, ws[0], act=act).clone()
        else:
            assert torch.equal(ops.gemm(a, ws[0], act=act), ref), "tile order changed the bits"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20 if M * N * K < 3e11 else 8
        e0.record()
        for i in range(iters):
            ops.gemm(a, ws[(i + 3) % ncopy], act=act)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        key = f"gm{gm}_us" if f"gm{gm}_us" not in row else f"gm{gm}_again_us"
        row[key] = round(us, 1)
    row["tflops_gm8"] = round(2.0 * M * N * K / row["gm8_us"] / 1e6, 1)
    res.append(row)
    print(json.dumps(row), flush=True)
    del ws
L.load().fo1_gemm_set_group_m(0)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_group_m_ab.json", "w"), indent=1)
