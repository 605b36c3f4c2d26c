"""Ablation of the LDS-DMA GEMM main loop (GPU box): where does the time go?  Results are INVALID numerically."""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops
SHAPES = [("llm_gateup", 515, 22016, 2048, 2), ("vit_qkv", 1564, 3840, 1280, 2), ("sq4096", 4096, 4096, 4096, 1), ("llm_qkv", 515, 2560, 2048, 3)]
MODES = [(0, "full"), (1, "no global loads"), (2, "no MFMA (loads + LDS reads)"), (4, "loads + barriers only"), (5, "barriers only"), (3, "LDS reads only")]
for name, M, N, K, tile in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.load().fo1_gemm_set_variant(2, tile)
    L.load().fo1_gemm_set_splitk(1)
    for bits, label in MODES:
        L.load().fo1_gemm_set_debug(bits)
        for _ in range(3):
            ops.gemm(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:12s} tile{tile} {label:30s} {us:8.1f} us  ({2.0*M*N*K/us/1e6:7.1f} TF-equivalent)", flush=True)
L.load().fo1_gemm_set_debug(0); L.load().fo1_gemm_set_variant(0, 0); L.load().fo1_gemm_set_splitk(0)
