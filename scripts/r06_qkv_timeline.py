"""Round 6: workgroup timeline of the fused q/k/v projection (fo1_qkv_proj_rope_bf16, ViT head-major form: M = 25 x 1564, N = 16 x 256, K = 1280)
next to the plain product of the same shape — fo1_gemm_set_debug bit 5 stamps s_memrealtime in waves 0 (a rotating wave) and 7 (V / pad columns only)
at kernel entry, first MFMA, end of the K loop, tile staged + barrier passed (slot 6), rotation done (slot 7) and end of the epilogue.
    FO1_AB=1 python scripts/r06_qkv_timeline.py out.json"""
import json
import os
import sys

os.environ.setdefault("FO1_AB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vlm_fo1_amd import lib as L, ops

lib = L.load()
M, H, K = 25 * 1564, 16, 1280
N = H * 256
torch.manual_seed(0)
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
cos = torch.rand(M, 40, device="cuda")
sin = torch.rand(M, 40, device="cuda")
Sp = (M + 63) // 64 * 64
vt = torch.zeros(H * 80, Sp, dtype=torch.bfloat16, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
tiles = -(-M // 256) * (N // 256)


def run(fused):
    fn = (lambda: ops.qkv_proj_rope(x, w, b, 1, H, H, cos, sin, None, 0, vt, out=out)) if fused else (lambda: ops.gemm(x, w, b, out=out))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 100.0
    stamps = torch.zeros(tiles * 2 * 8, dtype=torch.int64, device="cuda")
    L.check(lib.fo1_gemm_set_stamp_buffer(stamps.data_ptr()), "stamp buffer")
    L.check(lib.fo1_gemm_set_debug(32), "debug")
    fn()
    torch.cuda.synchronize()
    L.check(lib.fo1_gemm_set_debug(0), "debug")
    L.check(lib.fo1_gemm_set_stamp_buffer(None), "stamp buffer")
    st = stamps.cpu().numpy().reshape(tiles, 2, 8)
    t = st.astype(np.float64) / 100.0
    entry, first, kend, end, staged, rot = t[:, :, 0], t[:, :, 1], t[:, :, 2], t[:, :, 3], t[:, :, 6], t[:, :, 7]
    med = lambda a: round(float(np.median(a)), 2)
    row = dict(form="fused q/k/v epilogue" if fused else "plain bias epilogue", us_per_launch=round(us, 1), tiles=tiles, rounds=round(tiles / 256, 2),
               prologue_us=med(first - entry), k_loop_us=med(kend - first), epilogue_wave0_us=med((end - kend)[:, 0]), epilogue_wave7_us=med((end - kend)[:, 1]),
               workgroup_us=med(end.max(1) - entry.min(1)))
    if fused:
        row.update(wave0_stage_and_barrier_us=med((staged - kend)[:, 0]), wave0_rotation_us=med((rot - staged)[:, 0]), wave0_after_rotation_us=med((end - rot)[:, 0]),
                   wave7_stage_and_barrier_us=med((staged - kend)[:, 1]), wave7_vt_stores_us=med((end - rot)[:, 1]))
    print(row, flush=True)
    return row


rows = [run(False), run(True)]
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
