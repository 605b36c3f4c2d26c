#!/usr/bin/env python3
"""Would the hardware's per-32-element E8M0 block scales (v_mfma_scale_f32_32x32x64_f8f6f4's scale operands) bring the fp8 engine
to the 0.99 last-hidden bar?  CPU study on the arithmetic alone (torch float8_e4m3fn = the oracle's conversion, oracle/fp8_oracle.py): one linear
C = A W^T with (a) per-row fp32 scales = absmax / 448 (what fo1_gemm_fp8 does), (b) per-32-element power-of-two block scales along K for both
operands (MX-style: scale = 2^ceil(log2(block absmax / 448))), (c) both — on gaussian operands (the fixture weights) and on operands with outlier
channels (what trained checkpoints show).  Prints rms relative error and cosine per variant; writes profiles/r06_fp8_block_scale_study.json."""
import json
import os
import sys
import torch

torch.manual_seed(0)
F8 = torch.float8_e4m3fn


def q_rows(x):
    s = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
    return (x / s).to(F8).float() * s


def q_blocks(x, row_scale=False):
    M, K = x.shape
    if row_scale:
        s0 = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
        x = x / s0
    b = x.view(M, K // 32, 32)
    s = torch.exp2(torch.ceil(torch.log2(b.abs().amax(dim=2, keepdim=True).clamp_min(1e-30) / 448.0)))
    y = ((b / s).to(F8).float() * s).view(M, K)
    return y * s0 if row_scale else y


def study(name, A, W):
    ref = A @ W.t()
    out = {}
    for tag, qa, qw in (("per_row_fp32_scales (fo1_gemm_fp8)", q_rows(A), q_rows(W)),
                        ("per_32_block_e8m0_scales", q_blocks(A), q_blocks(W)),
                        ("per_row_then_block_scales", q_blocks(A, True), q_blocks(W, True))):
        c = qa @ qw.t()
        rel = ((c - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        cos = torch.nn.functional.cosine_similarity(c, ref, dim=1).min().item()
        out[tag] = dict(rms_rel_err=round(rel, 5), min_row_cos=round(cos, 6))
        print(f"{name:34s} {tag:40s} rms rel {rel:.4f}  min row cos {cos:.6f}")
    return out


res = {}
M, N, K = 512, 2048, 2048
A, W = torch.randn(M, K), torch.randn(N, K) * 0.02
res["gaussian (the fixture weights)"] = study("gaussian", A, W)
Ao = A.clone()
Ao[:, torch.randperm(K)[:8]] *= 60.0          # a few outlier channels in the activations, as trained LLMs have
res["activation outlier channels x60"] = study("outlier channels", Ao, W)
res["note"] = ("e4m3 keeps 3 mantissa bits: ~2.6-3 % rms error per product whatever the scale granularity; block scales help where a row's dynamic range "
               "is spent on outliers, which seeded gaussian fixtures do not have.  68 residual blocks of ~3 % noise are what puts last-hidden min-cos at 0.96-0.97 "
               "(profiles/r05_fp8_hires_metrics.json); per-32 block scales cannot move that on these weights.")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(res, open(os.path.join(root, "profiles", "r06_fp8_block_scale_study.json"), "w"), indent=1)
