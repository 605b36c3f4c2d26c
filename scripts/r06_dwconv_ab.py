"""Round 6: fo1_dwconv3x3_ln_bf16 at the DaViT stage shapes of the 25-image pass — sliding-window run form against the per-pixel form
(FO1_AB build: fo1_dwconv_ln_set_form).  Bitwise check, then hipEvent timing with the operands evicted from the caches between calls.
    FO1_AB=1 python scripts/r06_dwconv_ab.py out.json"""
import json
import os
import sys

os.environ.setdefault("FO1_AB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vlm_fo1_amd import lib as _L, ops

BF = torch.bfloat16


def timed(fn, flush, iters=20):
    ts = []
    for _ in range(iters):
        flush.add_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    out = {}
    L = _L.load()
    flush = torch.zeros(768 << 20, dtype=torch.uint8, device="cuda")
    torch.manual_seed(0)
    for (B, H, W, C) in [(25, 192, 192, 256), (25, 96, 96, 512), (25, 48, 48, 1024), (25, 24, 24, 2048), (1, 192, 192, 256), (1, 48, 48, 1024)]:
        x = torch.randn(B * H * W, C).to(BF).cuda()
        w9 = (torch.randn(9, C) * 0.2).to(BF).cuda()
        b = (torch.randn(C) * 0.1).to(BF).cuda()
        lw = (1 + 0.1 * torch.randn(C)).to(BF).cuda()
        lb = (0.1 * torch.randn(C)).to(BF).cuda()
        res = {}
        outs = {}
        for form in (0, 2):
            L.fo1_dwconv_ln_set_form(form)
            outs[form] = ops.dwconv3x3_res_ln(x, w9, b, H, W, lw, lb, 1e-5, batch=B)
            res["run" if form else "pixel"] = round(timed(lambda: ops.dwconv3x3_res_ln(x, w9, b, H, W, lw, lb, 1e-5, batch=B), flush), 2)
        res["bitwise"] = bool(torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]))
        res["algorithmic_MB"] = round(3 * B * H * W * C * 2 / 1e6, 1)
        res["run_GBps"] = round(res["algorithmic_MB"] / res["run"] * 1e3 / 1e3, 1)
        out[f"{B}x{H}x{W}x{C}"] = res
        print(f"{B}x{H}x{W}x{C}", res, flush=True)
    L.fo1_dwconv_ln_set_form(1)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
