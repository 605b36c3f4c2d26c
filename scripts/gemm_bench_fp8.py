"""fp8 (e4m3, v_mfma_scale 32x32x64) vs bf16 256x256 two-phase GEMM on the packed-pass shapes, cold weights (GPU box only).
usage: gemm_bench_fp8.py <out.json> [images]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlm_fo1_amd import lib as L, ops

B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S, Lp = 1564, 651
SHAPES = [
    ("vit_qkv", B * S, 3840, 1280), ("vit_proj", B * S, 1280, 1280), ("vit_gateup", B * S, 6912, 1280), ("vit_down", B * S, 1280, 3456),
    ("llm_qkv", B * Lp, 2560, 2048), ("llm_o", B * Lp, 2048, 2048), ("llm_gateup", B * Lp, 22016, 2048), ("llm_down", B * Lp, 2048, 11008),
    ("davit_s2_fc1", B * 1200, 4096, 1024), ("fpn3x3_l0", B * 25024, 512, 4608), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
]
res = []
for name, M, N, K in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ncopy = max(2, min(32, int(640e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(ncopy)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    aq, sa = ops.quantize_rows_fp8(a)
    wqs = [ops.Fp8Weight(*ops.quantize_rows_fp8(w)) for w in ws]
    iters = 20 if M * N * K < 3e11 else 6

    def timed(fn):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i + 3)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    t_bf = timed(lambda i: ops.gemm(a, ws[i % ncopy], out=out))
    t_f8 = timed(lambda i: ops.gemm_fp8(aq, sa, wqs[i % ncopy], out=out))
    t_q = timed(lambda i: ops.quantize_rows_fp8(a, aq, sa))
    fl = 2.0 * M * N * K
    r = dict(shape=name, M=M, N=N, K=K, bf16_us=round(t_bf * 1e3, 1), fp8_us=round(t_f8 * 1e3, 1), quantize_a_us=round(t_q * 1e3, 1),
             bf16_tflops=round(fl / t_bf / 1e9, 1), fp8_tflops=round(fl / t_f8 / 1e9, 1), fp8_incl_quant_tflops=round(fl / (t_f8 + t_q) / 1e9, 1))
    res.append(r)
    print(r, flush=True)
    del ws, wqs
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_bench_fp8.json", "w"), indent=1)
