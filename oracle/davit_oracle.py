"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement (plain torch fp32 on bf16-valued weights) of the auxiliary tower
  vlm_fo1/model/multimodal_encoder/davit/modeling_davit.py
    PreNorm :29-48, Mlp :51-69, DepthWiseConv2d :72-99, ConvEmbed :102-148, ChannelAttention :151-172,
    ChannelBlock :175-206, window_partition/reverse :208-222, WindowAttention :225-282,
    SpatialBlock :285-315, DaViT.forward_features :478-506
with the davit-large configuration of davit/configs.py:70-136.  Pinned in tests/test_oracle_davit.py
against the reference module imported in place (timm shim: DropPath / trunc_normal_ only).
State-dict keys are the reference module's.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

DAVIT_LARGE = dict(depths=(1, 1, 9, 1), dims=(256, 512, 1024, 2048), heads=(8, 16, 32, 64), groups=(8, 16, 32, 64),
                   patch_size=(7, 3, 3, 3), patch_stride=(4, 2, 2, 2), patch_padding=(3, 1, 1, 1),
                   patch_prenorm=(False, True, True, True), window=12)


def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"].float(), sd[p + "bias"].float(), eps)


def _dw(x, H, W, sd, p):
    C = x.shape[-1]
    y = F.conv2d(x.t().reshape(1, C, H, W), sd[p + "dw.weight"].float(), sd[p + "dw.bias"].float(), padding=1, groups=C)
    return x + y.flatten(2)[0].t()


def _mlp(x, sd, p):
    h = F.linear(_ln(x, sd, p + "norm."), sd[p + "fn.net.fc1.weight"].float(), sd[p + "fn.net.fc1.bias"].float())
    return x + F.linear(F.gelu(h), sd[p + "fn.net.fc2.weight"].float(), sd[p + "fn.net.fc2.bias"].float())


def _window_attn(x, H, W, sd, p, heads, ws):
    C = x.shape[-1]
    h = _ln(x, sd, p + "norm.").reshape(1, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    h = F.pad(h, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = h.shape[1:3]
    h = h.view(1, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    B_, N, _ = h.shape
    qkv = F.linear(h, sd[p + "fn.qkv.weight"].float(), sd[p + "fn.qkv.bias"].float())
    qkv = qkv.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (float(C // heads) ** -0.5), qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(B_, N, C)
    o = F.linear(o, sd[p + "fn.proj.weight"].float(), sd[p + "fn.proj.bias"].float())
    o = o.view(1, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(1, Hp, Wp, C)[:, :H, :W]
    return x + o.reshape(H * W, C)


def _channel_attn(x, sd, p, groups):
    N, C = x.shape
    qkv = F.linear(_ln(x, sd, p + "norm."), sd[p + "fn.qkv.weight"].float(), sd[p + "fn.qkv.bias"].float())
    qkv = qkv.reshape(1, N, 3, groups, C // groups).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (float(N) ** -0.5), qkv[1], qkv[2]
    att = (q.transpose(-1, -2) @ k).softmax(-1)
    o = (att @ v.transpose(-1, -2)).transpose(-1, -2).transpose(1, 2).reshape(N, C)
    return x + F.linear(o, sd[p + "fn.proj.weight"].float(), sd[p + "fn.proj.bias"].float())


def davit_forward(sd: Dict[str, torch.Tensor], img: torch.Tensor, cfg=DAVIT_LARGE) -> List[torch.Tensor]:
    """img [1,3,H,W] -> 4 token-major maps [(H_i*W_i, C_i)] plus their sizes."""
    x = img.float()
    H, W = x.shape[2:]
    outs, sizes = [], []
    tok = None
    for i in range(len(cfg["dims"])):
        pc = f"convs.{i}."
        if i == 0:
            y = F.conv2d(x, sd[pc + "proj.weight"].float(), sd[pc + "proj.bias"].float(), stride=cfg["patch_stride"][i],
                         padding=cfg["patch_padding"][i])
            H, W = y.shape[2:]
            tok = y.flatten(2)[0].t()
            tok = _ln(tok, sd, pc + "norm.")
        else:
            t = _ln(tok, sd, pc + "norm.") if cfg["patch_prenorm"][i] else tok
            Cp = t.shape[-1]
            y = F.conv2d(t.t().reshape(1, Cp, H, W), sd[pc + "proj.weight"].float(), sd[pc + "proj.bias"].float(),
                         stride=cfg["patch_stride"][i], padding=cfg["patch_padding"][i])
            H, W = y.shape[2:]
            tok = y.flatten(2)[0].t()
            if not cfg["patch_prenorm"][i]:
                tok = _ln(tok, sd, pc + "norm.")
        for j in range(cfg["depths"][i]):
            ps = f"blocks.{i}.{j}.spatial_block."
            tok = _dw(tok, H, W, sd, ps + "conv1.fn.")
            tok = _window_attn(tok, H, W, sd, ps + "window_attn.", cfg["heads"][i], cfg["window"])
            tok = _dw(tok, H, W, sd, ps + "conv2.fn.")
            tok = _mlp(tok, sd, ps + "ffn.")
            pch = f"blocks.{i}.{j}.channel_block."
            tok = _dw(tok, H, W, sd, pch + "conv1.fn.")
            tok = _channel_attn(tok, sd, pch + "channel_attn.", cfg["groups"][i])
            tok = _dw(tok, H, W, sd, pch + "conv2.fn.")
            tok = _mlp(tok, sd, pch + "ffn.")
        outs.append(tok)
        sizes.append((H, W))
    return outs, sizes


def random_davit_state(cfg=DAVIT_LARGE, seed=0, std=0.02):
    """Seeded bf16-valued random weights under the reference module's key names."""
    g = torch.Generator().manual_seed(seed)

    def w(*s, sc=std):
        return (torch.randn(*s, generator=g) * sc).bfloat16()

    def ln(sd, p, c):
        sd[p + "weight"] = (1 + 0.1 * torch.randn(c, generator=g)).bfloat16()
        sd[p + "bias"] = (0.05 * torch.randn(c, generator=g)).bfloat16()

    sd = {}
    prev = 3
    for i, c in enumerate(cfg["dims"]):
        k = cfg["patch_size"][i]
        sd[f"convs.{i}.proj.weight"] = w(c, prev, k, k, sc=0.05)
        sd[f"convs.{i}.proj.bias"] = w(c, sc=0.05)
        ln(sd, f"convs.{i}.norm.", prev if cfg["patch_prenorm"][i] else c)
        for j in range(cfg["depths"][i]):
            for blk, attn in (("spatial_block", "window_attn"), ("channel_block", "channel_attn")):
                p = f"blocks.{i}.{j}.{blk}."
                for cv in ("conv1", "conv2"):
                    sd[p + cv + ".fn.dw.weight"] = w(c, 1, 3, 3, sc=0.1)
                    sd[p + cv + ".fn.dw.bias"] = w(c, sc=0.05)
                ln(sd, p + attn + ".norm.", c)
                sd[p + attn + ".fn.qkv.weight"] = w(3 * c, c, sc=0.05)
                sd[p + attn + ".fn.qkv.bias"] = w(3 * c, sc=0.05)
                sd[p + attn + ".fn.proj.weight"] = w(c, c)
                sd[p + attn + ".fn.proj.bias"] = w(c, sc=0.05)
                ln(sd, p + "ffn.norm.", c)
                sd[p + "ffn.fn.net.fc1.weight"] = w(4 * c, c)
                sd[p + "ffn.fn.net.fc1.bias"] = w(4 * c, sc=0.05)
                sd[p + "ffn.fn.net.fc2.weight"] = w(c, 4 * c)
                sd[p + "ffn.fn.net.fc2.bias"] = w(c, sc=0.05)
        prev = c
    return sd
