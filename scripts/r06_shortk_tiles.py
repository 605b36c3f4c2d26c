"""Round 6: DaViT's short-K products (stages 1 and 2 of the 25-image pass: K = 256 / 512, M = 480 000 / 120 000 rows) over the tile shapes of
fo1_gemm_bf16 — is the 256 x 256 tile (one workgroup per CU, epilogue exposed) still the right choice when a tile has 4 - 8 K steps?
    FO1_AB=1 python scripts/r06_shortk_tiles.py out.json"""
import json
import os
import sys

os.environ.setdefault("FO1_AB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vlm_fo1_amd import lib as L, ops

SHAPES = [("s1_fc1_gelu", 480000, 1024, 256, ops.ACT_GELU, False), ("s1_fc2_res", 480000, 256, 1024, 0, True), ("s1_qkv", 480000, 768, 256, 0, False),
          ("s1_proj", 480000, 256, 256, 0, False), ("s2_fc1_gelu", 120000, 2048, 512, ops.ACT_GELU, False), ("s2_fc2_res", 120000, 512, 2048, 0, True),
          ("s2_qkv", 120000, 1536, 512, 0, False), ("s2_proj", 120000, 512, 512, 0, False)]
VARIANTS = [(0, 0), (2, 5), (2, 4), (2, 1), (3, 1), (3, 2)]
lib = L.load()
res = []
flush = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")
for name, M, N, K, act, has_res in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16() if has_res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = None
    for staging, tile in VARIANTS:
        lib.fo1_gemm_set_variant(staging, tile)
        lib.fo1_gemm_set_splitk(1)
        ts = []
        for i in range(7):
            flush.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(a, w, b, r, act, out=out); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        gb = (M * K + N * K + M * N * (2 if has_res else 1)) * 2 / 1e9
        row = dict(shape=name, M=M, N=N, K=K, staging=staging, tile=tile, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1), GBps=round(gb / us * 1e6, 0),
                   equals_auto=same)
        res.append(row)
        print(row, flush=True)
lib.fo1_gemm_set_variant(0, 0)
lib.fo1_gemm_set_splitk(0)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
