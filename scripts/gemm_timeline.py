"""Workgroup timeline of the 256x256 GEMM (GPU box only; test / bench build): fo1_gemm_set_debug bit 5 makes waves 0 and 7 of every workgroup
stamp s_memrealtime (100 MHz) at kernel entry, first MFMA, end of the K loop and end of the epilogue.  Per shape this prints, in microseconds:
prologue (entry -> first MFMA), K loop, epilogue, and the turnover gap between consecutive workgroups on the same CU (end of one -> entry
of the next), plus the shader clock the K loop ran at (s_memtime cycles / s_memrealtime).  usage: gemm_timeline.py [out.json] [images] [abl]
(abl: a main-loop ablation of scripts/gemm_loop_ablation.py for the plain / residual products, e.g. 1 = no DMA inside the loop — timing only)"""
import os
os.environ.setdefault("FO1_AB", "1")
import json, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vlm_fo1_amd import lib as L, ops

B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ABL = int(sys.argv[3]) if len(sys.argv) > 3 else 0
S, Lp = 1564, 651
SHAPES = [("vit_qkv", B * S, 3840, 1280, 0, True), ("vit_proj+res", B * S, 1280, 1280, 1, True), ("vit_gateup_swiglu", B * S, 6912, 1280, 3, True),
          ("llm_o+res", B * Lp, 2048, 2048, 1, False), ("llm_gateup_swiglu", B * Lp, 22016, 2048, 3, False), ("llm_down+res", B * Lp, 2048, 11008, 1, False)]
lib = L.load()
out = []
for name, M, N, K, mode, has_bias in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    n_out = N // 2 if mode == 3 else N
    c = torch.empty(M, n_out, dtype=torch.bfloat16, device="cuda")
    resid = torch.randn(M, N, device="cuda").bfloat16() if mode == 1 else None
    bias = torch.randn(N, device="cuda").bfloat16() if has_bias else None
    act = ops.ACT_SWIGLU16 if mode == 3 else ops.ACT_NONE
    tiles = -(-M // 256) * -(-N // 256)
    stamps = torch.zeros(tiles * 2 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, bias, resid, act, out=c)
    L.check(lib.fo1_gemm_set_stamp_buffer(stamps.data_ptr()), "stamp buffer")
    L.check(lib.fo1_gemm_set_debug(32 | (ABL << 6)), "debug")
    ops.gemm(a, w, bias, resid, act, out=c)
    torch.cuda.synchronize()
    L.check(lib.fo1_gemm_set_debug(0), "debug")
    L.check(lib.fo1_gemm_set_stamp_buffer(None), "stamp buffer")
    st = stamps.cpu().numpy().reshape(tiles, 2, 8)
    t = st[:, :, :4].astype(np.float64) / 100.0            # us
    t0 = t[:, :, 0].min()
    entry, first, kend, end = (t[:, :, i] for i in range(4))
    hw, xcc = st[:, 0, 4] & 0xFFFFFFFF, st[:, 0, 5] & 0xF
    cyc = ((st[:, :, 5] >> 32) - (st[:, :, 4] >> 32)) & 0xFFFFFFFF          # shader cycles of the K loop (s_memtime, low 32 bits)
    ghz = cyc.astype(np.float64) / np.maximum((kend - first) * 1e3, 1.0)
    cu = (xcc.astype(np.int64) << 16) | (hw & 0xFF00)      # XCC | SE, SH, CU bits of HW_ID (bits 8..15)
    wg_entry, wg_end = entry.min(1), end.max(1)
    gaps = []
    for cid in np.unique(cu):
        idx = np.where(cu == cid)[0]
        idx = idx[np.argsort(wg_entry[idx])]
        gaps += list(wg_entry[idx[1:]] - wg_end[idx[:-1]])
    row = dict(shape=name, ablation=ABL, M=M, N=N, K=K, tiles=int(tiles), cus=int(len(np.unique(cu))),
               kernel_us=round(float(wg_end.max() - t0), 1),
               prologue_us=round(float(np.median(first - entry)), 2), k_loop_us=round(float(np.median(kend - first)), 2),
               epilogue_early_half_us=round(float(np.median((end - kend)[:, 0])), 2), epilogue_late_half_us=round(float(np.median((end - kend)[:, 1])), 2),
               epilogue_convert_to_lds_us=(round(float(np.median((st[:, 0, 6].astype(np.float64) / 100.0) - kend[:, 0])), 2) if mode != 3 else None),
               late_half_lag_us=round(float(np.median(kend[:, 1] - kend[:, 0])), 2),
               turnover_gap_us=round(float(np.median(gaps)), 2) if gaps else None, turnover_gap_p90_us=round(float(np.percentile(gaps, 90)), 2) if gaps else None,
               k_loop_clock_ghz=round(float(np.median(ghz)), 3), k_loop_cycles_per_k_tile=round(float(np.median(cyc)) / (K // 64), 1),
               k_tiles=K // 64, us_per_k_tile=round(float(np.median(kend - first)) / (K // 64), 3))
    out.append(row)
    print(row, flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_timeline.json", "w"), indent=1)
