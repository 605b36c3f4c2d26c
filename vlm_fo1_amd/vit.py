"""Qwen2.5-VL vision tower on the MI355X engine.

Mirrors the reference's `Qwen2_5_VlVisionTower.forward` + monkey-patched `custom_forward` +
`VisionFeaturesGather.extract_multi_level_features` (qwen2_5_vl_encoder.py:37-158,228-257): patch
embed as a GEMM, window re-order, 32 blocks (windowed / full varlen attention), merger, and the
hidden states captured after the full-attention blocks — written straight into RASTER order
token-major maps [gh*gw, 1280] (the un-window/un-merge shuffle of :57-70 is folded into one row
gather), which is exactly the layout the HFRE kernel reads.

All index bookkeeping the reference does on device with .item()/.tolist() syncs (get_window_index,
rot_pos_emb, cu_seqlens; modeling_qwen2_5_vl.py:436-504) is computed once per grid shape on the host
and cached.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class ViTConfig:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    out_hidden_size: int = 2048
    patch_size: int = 14
    spatial_merge_size: int = 2
    temporal_patch_size: int = 2
    in_channels: int = 3
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = (7, 15, 23, 31)


def _round_up(x, m):
    return (x + m - 1) // m * m


class GridPlan:
    """Host-side index plan for one patch grid (gh, gw)."""

    def __init__(self, gh: int, gw: int, cfg: ViTConfig, device):
        m = cfg.spatial_merge_size
        unit = m * m
        S = gh * gw
        vmw = cfg.window_size // m // cfg.patch_size
        lh, lw = gh // m, gw // m
        index = torch.arange(lh * lw).reshape(1, lh, lw)
        pad_h, pad_w = vmw - lh % vmw, vmw - lw % vmw
        nwh, nww = (lh + pad_h) // vmw, (lw + pad_w) // vmw
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(1, nwh, vmw, nww, vmw).permute(0, 1, 3, 2, 4).reshape(1, nwh * nww, vmw, vmw)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        widx = ip.reshape(-1)
        widx = widx[widx != -100]                       # window order -> merge-unit index
        cu = [0] + (seqlens.cumsum(0) * unit).tolist()
        cu = sorted(set(cu))
        # row permutation: window-order row r reads merge-block-order row perm[r]
        perm = (widx[:, None] * unit + torch.arange(unit)[None, :]).reshape(-1)
        inv_unit = torch.argsort(widx)                   # merge unit -> position in window order
        # raster (y, x) -> merge-block row -> window-order row
        ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        mrow = ((ys // m) * lw + (xs // m)) * unit + (ys % m) * m + (xs % m)
        raster_to_win = (inv_unit[mrow // unit] * unit + mrow % unit).reshape(-1)
        # 2-D rope angles in merge-block order, then window order
        hd = cfg.hidden_size // cfg.num_heads
        dim = hd // 2
        inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
        full = torch.outer(torch.arange(max(gh, gw), dtype=torch.float), inv)
        hpos = torch.arange(gh).unsqueeze(1).expand(-1, gw).reshape(lh, m, lw, m).permute(0, 2, 1, 3).flatten()
        wpos = torch.arange(gw).unsqueeze(0).expand(gh, -1).reshape(lh, m, lw, m).permute(0, 2, 1, 3).flatten()
        fr = full[torch.stack([hpos, wpos], -1)].flatten(1)[perm]   # [S, hd/2]
        self.gh, self.gw, self.S = gh, gw, S
        z = torch.zeros(S, dtype=torch.int32)
        self.plan_in = torch.stack([z, perm.to(torch.int32)], 1).contiguous().to(device)
        self.plan_raster = torch.stack([z, raster_to_win.to(torch.int32)], 1).contiguous().to(device)
        zu = torch.zeros(S // unit, dtype=torch.int32)
        self.plan_tokens = torch.stack([zu, inv_unit.to(torch.int32)], 1).contiguous().to(device)
        self.cos = fr.cos().contiguous().to(device)
        self.sin = fr.sin().contiguous().to(device)
        wins = [(a, b) for a, b in zip(cu[:-1], cu[1:])]
        self.items_win = ops.make_items(wins, device, block=ops.pick_q_block(wins, cfg.num_heads))
        self.items_win.single_tile = ops.single_tile_items(wins, cfg.hidden_size // cfg.num_heads)      # -> fo1_attention_windows_bf16
        self.items_full = ops.make_items([(0, S)], device, block=ops.pick_q_block([(0, S)], cfg.num_heads, cfg.hidden_size // cfg.num_heads))
        self.cu_window = cu
        self.Sp = _round_up(S, 64)


class BatchPlan:
    """Index plan for several images packed row-wise into ONE pass: image i owns rows [row0[i], row0[i] + S_i) of
    every activation; windows / full-attention segments, 2-D rope angles and the gather plans are the per-image plans with row
    offsets added.  Grids may differ from image to image (every op of the ViT is row- or segment-wise)."""

    def __init__(self, plans: Sequence[GridPlan], cfg: ViTConfig, device):
        unit = cfg.spatial_merge_size ** 2
        self.grids = [(g.gh, g.gw) for g in plans]
        self.row0, off = [], 0
        for g in plans:
            self.row0.append(off)
            off += g.S
        self.S = off
        self.Sp = _round_up(off, 64)

        def cat_plan(parts, offs):
            out = []
            for t, o in zip(parts, offs):
                t = t.clone()
                t[:, 1] += o
                out.append(t)
            return torch.cat(out, 0).contiguous()

        self.plan_in = cat_plan([g.plan_in for g in plans], self.row0)
        self.plan_raster = cat_plan([g.plan_raster for g in plans], self.row0)
        self.plan_tokens = cat_plan([g.plan_tokens for g in plans], [o // unit for o in self.row0])
        self.cos = torch.cat([g.cos for g in plans], 0).contiguous()
        self.sin = torch.cat([g.sin for g in plans], 0).contiguous()
        wins, fulls = [], []
        for g, o in zip(plans, self.row0):
            wins += [(a + o, b + o) for a, b in zip(g.cu_window[:-1], g.cu_window[1:])]
            fulls.append((o, o + g.S))
        self.cu_window = None
        self.win_segments, self.full_segments = wins, fulls
        self.items_win = ops.make_items(wins, device, block=ops.pick_q_block(wins, cfg.num_heads))
        self.items_win.single_tile = ops.single_tile_items(wins, cfg.hidden_size // cfg.num_heads)
        self.items_full = ops.make_items(fulls, device, block=ops.pick_q_block(fulls, cfg.num_heads, cfg.hidden_size // cfg.num_heads))
        self.gh = self.gw = None


class QwenViT:
    def __init__(self, cfg: ViTConfig, state: Dict[str, torch.Tensor], device):
        self.cfg = cfg
        self.dev = torch.device(device)
        bf = torch.bfloat16
        d, ff = cfg.hidden_size, cfg.intermediate_size
        self.ffp = _round_up(ff, 64)  # 3420 -> 3456: zero rows/cols are exact and make K % 64 == 0
        # fused q/k/v epilogue (round 5): RoPE + V^T in the q/k/v GEMM's epilogue for passes large enough for the 256 x 256 GEMM kernel (ops.qkv_fused_for) — needs head dim 80
        # (three heads' worth of columns + pad = one 256-wide tile) and a second, head-major copy of the qkv weights (315 MB at depth 32)
        self._fused_qkv = ops.qkv_fused_enabled() and d // cfg.num_heads == 80 and d % 64 == 0

        def dv(t):
            return t.to(device=self.dev, dtype=bf).contiguous()

        def padrows(t, n):
            out = torch.zeros(n, *t.shape[1:], dtype=t.dtype)
            out[:t.shape[0]] = t
            return out

        pw = state["patch_embed.proj.weight"].reshape(d, -1)
        self.k_in = pw.shape[1]
        self.k_in_p = _round_up(self.k_in, 64)  # 1176 -> 1216
        pwp = torch.zeros(d, self.k_in_p, dtype=pw.dtype)
        pwp[:, :self.k_in] = pw
        self.patch_w = dv(pwp)
        self.blocks = []
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            wd = torch.zeros(d, self.ffp, dtype=state[p + "mlp.down_proj.weight"].dtype)
            wd[:, :ff] = state[p + "mlp.down_proj.weight"]
            self.blocks.append(dict(
                n1=dv(state[p + "norm1.weight"]), n2=dv(state[p + "norm2.weight"]),
                wqkv=dv(state[p + "attn.qkv.weight"]), bqkv=dv(state[p + "attn.qkv.bias"]),
                # head-major copy for the fused q/k/v epilogue (ops.qkv_proj_rope mode 1): per head [q | k | v | 16 zero rows] = one 256-column tile
                wqkv_hm=(dv(ops.head_major_qkv(state[p + "attn.qkv.weight"], cfg.num_heads, d // cfg.num_heads)) if self._fused_qkv else None),
                bqkv_hm=(dv(ops.head_major_qkv(state[p + "attn.qkv.bias"], cfg.num_heads, d // cfg.num_heads)) if self._fused_qkv else None),
                wo=dv(state[p + "attn.proj.weight"]), bo=dv(state[p + "attn.proj.bias"]),
                wgu=dv(ops.interleave_gate_up(padrows(state[p + "mlp.gate_proj.weight"], self.ffp), padrows(state[p + "mlp.up_proj.weight"], self.ffp))),
                bgu=dv(ops.interleave_gate_up(padrows(state[p + "mlp.gate_proj.bias"], self.ffp), padrows(state[p + "mlp.up_proj.bias"], self.ffp))),
                wd=dv(wd), bd=dv(state[p + "mlp.down_proj.bias"]),
            ))
        self.ln_q = dv(state["merger.ln_q.weight"])
        self.m0w, self.m0b = dv(state["merger.mlp.0.weight"]), dv(state["merger.mlp.0.bias"])
        self.m2w, self.m2b = dv(state["merger.mlp.2.weight"]), dv(state["merger.mlp.2.bias"])
        self._plans: Dict[Tuple[int, int], GridPlan] = {}
        self._bplans: Dict[tuple, BatchPlan] = {}

    def plan(self, gh: int, gw: int) -> GridPlan:
        key = (gh, gw)
        if key not in self._plans:
            self._plans[key] = GridPlan(gh, gw, self.cfg, self.dev)
        return self._plans[key]

    def batch_plan(self, grids: Sequence[Tuple[int, int]]) -> BatchPlan:
        key = tuple((int(a), int(b)) for a, b in grids)
        if key not in self._bplans:
            if len(self._bplans) >= 64:
                self._bplans.pop(next(iter(self._bplans)))
            self._bplans[key] = BatchPlan([self.plan(a, b) for a, b in key], self.cfg, self.dev)
        return self._bplans[key]

    def forward_batch(self, pixel_values: torch.Tensor, grids: Sequence[Tuple[int, int]], capture: str = "all"):
        """Several images in one pass: pixel_values [sum S_i, 1176] (the images' patch rows concatenated), grids [(gh_i, gw_i)].
        Returns (image tokens [sum S_i / 4, out_hidden] — image i at rows [row0_i / 4, ...), feature maps [sum S_i, 1280] raster
        per image at rows [row0_i, row0_i + S_i), plan)."""
        bp = self.batch_plan(grids)
        tokens, feats = self._forward(pixel_values, bp, capture)
        return tokens, feats, bp

    def forward(self, pixel_values: torch.Tensor, gh: int, gw: int, capture: str = "all"):
        """pixel_values [S, 1176] (device, bf16, merge-block order as the HF processor emits them).
        Returns (image_tokens [S/4, out_hidden] raster-merged order,
                 feature maps: list of [gh*gw, 1280] raster token-major (all full-attention blocks, or
                 only the last one when capture == "last" — what the SimpleFPN variant consumes; none when capture == "none"))."""
        g = self.plan(gh, gw)
        if pixel_values.shape != (g.S, self.k_in):
            raise ValueError(f"pixel_values {tuple(pixel_values.shape)} does not match grid {gh}x{gw} (expected [{g.S}, {self.k_in}])")
        return self._forward(pixel_values, g, capture)

    def _forward(self, pixel_values: torch.Tensor, g, capture: str):
        c = self.cfg
        S, d, H = g.S, c.hidden_size, c.num_heads
        hd = d // H
        if pixel_values.shape != (S, self.k_in):
            raise ValueError(f"pixel_values {tuple(pixel_values.shape)} does not match the plan (expected [{S}, {self.k_in}])")
        from . import stage_abi
        if stage_abi.enabled():      # the same launches, sequenced by fo1_vit_forward (csrc/stages.hip)
            return stage_abi.vit_stage(self).forward(pixel_values, g, capture)
        pix = pixel_values.to(torch.bfloat16)
        # window re-order folded into the patch-embed input gather; pad K 1176 -> 1216 (zeros)
        xin = ops.zero_framed(("vit_xin", S, self.k_in_p, self.k_in), S, self.k_in_p, self.dev)      # (the gather below writes columns [0, k_in) only)
        ops.gather_rows_into(g.plan_in, self.k_in, pix, out=xin)
        x = ops.gemm(xin, self.patch_w)
        vt = ops.zero_framed(("vit_vt", d, g.Sp, S), d, g.Sp, self.dev)  # V^T scratch, reused by every block; columns [S, Sp) are never written
        scale = 1.0 / math.sqrt(hd)
        if g.cu_window is not None:
            win_seg, full_seg = list(zip(g.cu_window[:-1], g.cu_window[1:])), [(0, S)]
        else:
            win_seg, full_seg = g.win_segments, g.full_segments
        fl_win = 4.0 * d * sum((b - a) ** 2 for a, b in win_seg)
        fl_full = 4.0 * d * sum((b - a) ** 2 for a, b in full_seg)
        feats: List[torch.Tensor] = []
        fused_qkv = self._fused_qkv and ops.qkv_fused_for(S, 3 * d, d) and not ops.fp8_routed(self.blocks[0]["wqkv"], S)
        for i, w in enumerate(self.blocks):
            full = i in c.fullatt_block_indexes
            if fused_qkv:
                # q/k/v projection with 2-D RoPE + V -> V^T in the GEMM's epilogue (one launch, no second pass over [S, 3 d]); head-major columns
                qkv = ops.qkv_proj_rope(ops.rmsnorm(x, w["n1"], 1e-6), w["wqkv_hm"], w["bqkv_hm"], 1, H, H, g.cos, g.sin, None, 0, vt)
                if not full and getattr(g.items_win, "single_tile", False):
                    att = ops.attention_windows(qkv, qkv[:, hd:], vt, g.items_win, H, hd, scale, flops=fl_win, qk_head_stride=256)
                else:
                    att = ops.attention(qkv, qkv[:, hd:], vt, g.items_full if full else g.items_win, H, H, hd, scale, False,
                                        flops=fl_full if full else fl_win, qk_head_stride=256)
            else:
                qkv = ops.norm_linear(x, w["n1"], 1e-6, w["wqkv"], w["bqkv"])
                ops.qkv_post_vit(qkv, H, hd, g.cos, g.sin, vt)   # 2-D RoPE on q/k + V -> V^T, one launch
                if not full and getattr(g.items_win, "single_tile", False):
                    att = ops.attention_windows(qkv[:, :d], qkv[:, d:2 * d], vt, g.items_win, H, hd, scale, flops=fl_win)
                else:
                    att = ops.attention(qkv[:, :d], qkv[:, d:2 * d], vt, g.items_full if full else g.items_win, H, H, hd, scale, False,
                                        flops=fl_full if full else fl_win)
            x = ops.gemm(att, w["wo"], w["bo"], residual=x)
            a = ops.norm_linear(x, w["n2"], 1e-6, w["wgu"], w["bgu"], act=ops.ACT_SWIGLU16)
            x = ops.gemm(a, w["wd"], w["bd"], residual=x)
            if full and capture != "none" and (capture == "all" or i == c.fullatt_block_indexes[-1]):
                feats.append(ops.gather_rows(g.plan_raster, d, x))
        u = c.spatial_merge_size ** 2
        m = ops.rmsnorm(x, self.ln_q, 1e-6).view(S // u, u * d)
        m = ops.gemm(m, self.m0w, self.m0b, act=ops.ACT_GELU)
        m = ops.gemm(m, self.m2w, self.m2b)
        tokens = ops.gather_rows(g.plan_tokens, c.out_hidden_size, m)
        return tokens, feats
