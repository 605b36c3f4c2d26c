"""Where the per-tile overhead of the 256x256 GEMM goes (GPU box only): normal / non-temporal epilogue stores / no epilogue stores /
one K tile only (prologue + epilogue), cold weights.  usage: gemm_t0_study.py <out.json> [images]"""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
S, Lp = 1564, 651
SHAPES = [("vit_qkv", B * S, 3840, 1280, 0), ("vit_proj+res", B * S, 1280, 1280, 1), ("vit_gateup_swiglu", B * S, 6912, 1280, 3), ("vit_down+res", B * S, 1280, 3456, 1),
          ("llm_o+res", B * Lp, 2048, 2048, 1), ("llm_gateup_swiglu", B * Lp, 22016, 2048, 3), ("davit_s0_fc1_gelu", B * 19200, 1024, 256, 2), ("sq8192", 8192, 8192, 8192, 0)]
VARIANTS = [("warmup", 1, 0), ("normal", 1, 0), ("persistent", 1 | 4, 0), ("nt_stores", 1 | 8, 0), ("no_stores", 1, 8), ("one_k_tile", 1, 16), ("one_k_tile_no_stores", 1, 24), ("persistent_again", 1 | 4, 0), ("normal_again", 1, 0)]
lib = L.load()
res = []
for name, M, N, K, mode in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ncopy = max(2, min(32, int(640e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    n_out = N // 2 if mode == 3 else N
    out = torch.empty(M, n_out, dtype=torch.bfloat16, device="cuda")
    resid = torch.randn(M, N, device="cuda").bfloat16() if mode == 1 else None
    bias = torch.randn(N, device="cuda").bfloat16()
    act = {0: ops.ACT_NONE, 1: ops.ACT_NONE, 2: ops.ACT_GELU, 3: ops.ACT_SWIGLU16}[mode]
    row = dict(shape=name, M=M, N=N, K=K)
    for label, sched, dbg in VARIANTS:
        lib.fo1_gemm_set_big_schedule(sched)
        lib.fo1_gemm_set_debug(dbg)
        for i in range(3):
            ops.gemm(a, ws[i % ncopy], bias, resid, act, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20 if M * N * K < 3e11 else 6
        e0.record()
        for i in range(iters):
            ops.gemm(a, ws[(i + 3) % ncopy], bias, resid, act, out=out)
        e1.record()
        torch.cuda.synchronize()
        row[label + "_us"] = round(e0.elapsed_time(e1) / iters * 1e3, 1)
    tiles = -(-M // 256) * -(-N // 256)
    row["tiles"] = tiles
    row["rounds"] = -(-tiles // 256)
    row["tflops_normal"] = round(2.0 * M * N * K / row["normal_us"] / 1e6, 1)
    res.append(row)
    print(row, flush=True)
    del ws
lib.fo1_gemm_set_debug(0)
lib.fo1_gemm_set_big_schedule(1)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_t0_study.json", "w"), indent=1)
