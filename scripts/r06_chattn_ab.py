"""Round 6: fo1_channel_attention_bf16 at the DaViT stage shapes of the 25-image pass — matrix-core Gram / product kernels against the fp32
FMA kernels (FO1_AB build: fo1_channel_attention_set_impl).  us per call (three launches), operands evicted between calls.
    FO1_AB=1 python scripts/r06_chattn_ab.py out.json"""
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
    for (B, N, C) in [(25, 19200, 256), (25, 4800, 512), (25, 1200, 1024), (25, 300, 2048), (1, 1200, 1024)]:
        qkv = torch.randn(B * N, 3 * C).to(BF).cuda()
        res, outs = {}, {}
        for impl in (0, 1):
            L.fo1_channel_attention_set_impl(impl)
            outs[impl] = ops.channel_attention(qkv, C, batch=B)
            res["mfma" if impl else "fma"] = round(timed(lambda: ops.channel_attention(qkv, C, batch=B), flush), 2)
        d = (outs[0].float() - outs[1].float()).abs()
        res["fraction_equal"] = round(float((d == 0).float().mean()), 5)
        res["max_abs_diff"] = float(d.max())
        res["algorithmic_MB"] = round(4 * B * N * C * 2 / 1e6, 1)
        out[f"{B}x{N}x{C}"] = res
        print(f"{B}x{N}x{C}", res, flush=True)
    L.fo1_channel_attention_set_impl(1)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
