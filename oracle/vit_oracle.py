"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement (plain torch fp32 on bf16-valued weights) of the primary vision tower path:
  vlm_fo1/model/multimodal_encoder/qwen2_5_vl_encoder.py
    custom_forward                 :86-158   (patch embed, window re-order, 32 blocks, captures, merger)
    extract_multi_level_features   :37-80    (un-window, un-merge -> 4 x [1,1280,gh,gw])
  vlm_fo1/model/multimodal_encoder/qwen2_5_vl/modeling_qwen2_5_vl.py
    Qwen2_5_VisionPatchEmbed :88-111, Qwen2RMSNorm :126-140, Qwen2_5_VLPatchMerger :146-159,
    apply_rotary_pos_emb_vision :219-230, attention :283-323, block :333-357,
    rot_pos_emb :436-463, get_window_index :465-504
Pinned in tests/test_oracle_vit.py against the vendored modules imported in place from
/root/reference (the ViT half runs under the installed transformers with sdpa).
State-dict keys = the checkpoint's (`blocks.{i}.attn.qkv.weight`, `merger.mlp.0.weight`, ...).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


def rmsnorm(x, w, eps=1e-6):
    v = x.float()
    return w.float() * (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps))


def window_index(gh: int, gw: int, merge: int = 2, window_size: int = 112, patch: int = 14):
    """-> (window_index over merge units [n_units], cu_window_seqlens in patch tokens (unique_consecutive'd))
    for one image, t = 1 (get_window_index :465-504 + custom_forward :104-109)."""
    vmw = window_size // merge // patch
    lh, lw = gh // merge, gw // merge
    index = torch.arange(lh * lw).reshape(1, lh, lw)
    pad_h = vmw - lh % vmw
    pad_w = vmw - lw % vmw
    nwh, nww = (lh + pad_h) // vmw, (lw + pad_w) // vmw
    ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
    ip = ip.reshape(1, nwh, vmw, nww, vmw).permute(0, 1, 3, 2, 4).reshape(1, nwh * nww, vmw, vmw)
    seqlens = (ip != -100).sum([2, 3]).reshape(-1)
    ip = ip.reshape(-1)
    widx = ip[ip != -100]
    cu = [0] + (seqlens.cumsum(0) * merge * merge).tolist()
    cu = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
    return widx, cu


def rot_pos_ids(gh: int, gw: int, merge: int = 2):
    """(h, w) position of every patch in the processor's merge-block order (rot_pos_emb :436-458)."""
    h = torch.arange(gh).unsqueeze(1).expand(-1, gw)
    h = h.reshape(gh // merge, merge, gw // merge, merge).permute(0, 2, 1, 3).flatten()
    w = torch.arange(gw).unsqueeze(0).expand(gh, -1)
    w = w.reshape(gh // merge, merge, gw // merge, merge).permute(0, 2, 1, 3).flatten()
    return torch.stack([h, w], dim=-1)


def rot_freqs(gh: int, gw: int, head_dim: int, theta: float = 10000.0):
    """[S, head_dim/2] angles in merge-block order (:114-123, :459-463)."""
    dim = head_dim // 2
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    seq = torch.arange(max(gh, gw), dtype=torch.float)
    full = torch.outer(seq, inv)             # [max, dim/2]
    return full[rot_pos_ids(gh, gw)].flatten(1)  # [S, dim]


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def vit_forward(sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, gh: int, gw: int, *, depth: int, n_heads: int,
                fullatt: Sequence[int], merge: int = 2, window_size: int = 112, patch: int = 14):
    """pixel_values [S, C*T*P*P] in merge-block order.
    Returns (image_tokens [S/4, out_dim] in raster-merged order, [feature maps [gh*gw, d] in raster order
    for every full-attention block])."""
    S = gh * gw
    unit = merge * merge
    d = sd["patch_embed.proj.weight"].shape[0]
    hd = d // n_heads
    x = F.linear(pixel_values.float(), sd["patch_embed.proj.weight"].float().reshape(d, -1))
    widx, cu_win = window_index(gh, gw, merge, window_size, patch)
    fr = rot_freqs(gh, gw, hd)
    x = x.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
    fr = fr.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
    cu_full = torch.tensor([0, S], dtype=torch.int32)
    feats = []
    for i in range(depth):
        p = f"blocks.{i}."
        cu = cu_full if i in fullatt else cu_win
        r = rmsnorm(x, sd[p + "norm1.weight"])
        qkv = F.linear(r, sd[p + "attn.qkv.weight"].float(), sd[p + "attn.qkv.bias"].float())
        q, k, v = qkv.reshape(S, 3, n_heads, hd).permute(1, 0, 2, 3).unbind(0)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        o = torch.zeros(S, n_heads, hd)
        for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
            att = torch.einsum("qhd,khd->hqk", q[a:b], k[a:b]) / math.sqrt(hd)
            o[a:b] = torch.einsum("hqk,khd->qhd", att.softmax(-1), v[a:b])
        x = x + F.linear(o.reshape(S, d), sd[p + "attn.proj.weight"].float(), sd[p + "attn.proj.bias"].float())
        r = rmsnorm(x, sd[p + "norm2.weight"])
        g = F.linear(r, sd[p + "mlp.gate_proj.weight"].float(), sd[p + "mlp.gate_proj.bias"].float())
        u = F.linear(r, sd[p + "mlp.up_proj.weight"].float(), sd[p + "mlp.up_proj.bias"].float())
        x = x + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"].float(), sd[p + "mlp.down_proj.bias"].float())
        if i in fullatt:
            feats.append(x.clone())
    m = rmsnorm(x, sd["merger.ln_q.weight"]).reshape(S // unit, unit * d)
    m = F.linear(m, sd["merger.mlp.0.weight"].float(), sd["merger.mlp.0.bias"].float())
    m = F.linear(F.gelu(m), sd["merger.mlp.2.weight"].float(), sd["merger.mlp.2.bias"].float())
    rev = torch.argsort(widx)
    tokens = m[rev]
    # extract_multi_level_features (:37-80): undo the window order, then the 2x2 merge-block order
    maps = []
    mh, mw = gh // merge, gw // merge
    for f in feats:
        f = f.reshape(S // unit, unit, -1)[rev].reshape(mh, mw, merge, merge, -1).permute(0, 2, 1, 3, 4).reshape(gh * gw, -1)
        maps.append(f)
    return tokens, maps


def random_vit_state(depth, d, n_heads, d_ff, out_dim, in_dim=1176, merge=2, seed=0, std=0.02):
    g = torch.Generator().manual_seed(seed)

    def w(*s, sc=std):
        return (torch.randn(*s, generator=g) * sc).bfloat16()

    sd = {"patch_embed.proj.weight": w(d, 3, 2, 14, 14) if in_dim == 1176 else w(d, in_dim)}
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()
        sd[p + "norm2.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()
        sd[p + "attn.qkv.weight"] = w(3 * d, d)
        sd[p + "attn.qkv.bias"] = w(3 * d, sc=0.1)
        sd[p + "attn.proj.weight"] = w(d, d)
        sd[p + "attn.proj.bias"] = w(d, sc=0.05)
        sd[p + "mlp.gate_proj.weight"] = w(d_ff, d)
        sd[p + "mlp.gate_proj.bias"] = w(d_ff, sc=0.05)
        sd[p + "mlp.up_proj.weight"] = w(d_ff, d)
        sd[p + "mlp.up_proj.bias"] = w(d_ff, sc=0.05)
        sd[p + "mlp.down_proj.weight"] = w(d, d_ff)
        sd[p + "mlp.down_proj.bias"] = w(d, sc=0.05)
    u = merge * merge
    sd["merger.ln_q.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()
    sd["merger.mlp.0.weight"] = w(u * d, u * d)
    sd["merger.mlp.0.bias"] = w(u * d, sc=0.05)
    sd["merger.mlp.2.weight"] = w(out_dim, u * d)
    sd["merger.mlp.2.bias"] = w(out_dim, sc=0.05)
    return sd
