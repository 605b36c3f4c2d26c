"""DaViT-L auxiliary tower on the MI355X engine.

Mirrors `DavitVisionTower.forward` -> `DaViT.forward_features`
(davit_aux_encoder.py:54-69, davit/modeling_davit.py:478-506) and returns the four stage outputs as
token-major [H_i*W_i, C_i] bf16 maps — the memory layout the reference's NCHW *views* already have
(modeling_davit.py:493) and the one the HFRE kernel reads.

Every conv is a GEMM: the 7x7/s4 and 3x3/s2 ConvEmbeds run as im2col + fo1_gemm_bf16 with the weights
re-laid out [Cout][ky][kx][Cin] at load time; depthwise 3x3 convs, window partition/reverse and the
channel attention are dedicated HBM-bound kernels (vision_ops.hip)."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import os

import torch

from . import ops

DAVIT_LARGE = dict(depths=(1, 1, 9, 1), dims=(256, 512, 1024, 2048), heads=(8, 16, 32, 64), groups=(8, 16, 32, 64),
                   patch_size=(7, 3, 3, 3), patch_stride=(4, 2, 2, 2), patch_padding=(3, 1, 1, 1),
                   patch_prenorm=(False, True, True, True), window=12)  # davit/configs.py:70-136


def _round_up(x, m):
    return (x + m - 1) // m * m


class DaViT:
    def __init__(self, state: Dict[str, torch.Tensor], device, cfg=DAVIT_LARGE):
        self.cfg = cfg
        self.dev = torch.device(device)
        bf = torch.bfloat16

        def dv(t):
            return t.to(device=self.dev, dtype=bf).contiguous()

        self.convs = []
        prev = 3
        for i, c in enumerate(cfg["dims"]):
            w = state[f"convs.{i}.proj.weight"]  # [Cout, Cin, k, k]
            k = cfg["patch_size"][i]
            cin_p = 8 if i == 0 else prev           # stage 0: image staged as [H*W, 8] (channels 3..7 zero)
            wp = torch.zeros(c, k, k, cin_p, dtype=w.dtype)
            wp[..., :w.shape[1]] = w.permute(0, 2, 3, 1)
            K = k * k * cin_p
            Kp = _round_up(K, 64)
            wg = torch.zeros(c, Kp, dtype=w.dtype)
            wg[:, :K] = wp.reshape(c, K)
            self.convs.append(dict(w=dv(wg), b=dv(state[f"convs.{i}.proj.bias"]), K=K, Kp=Kp,
                                   nw=dv(state[f"convs.{i}.norm.weight"]), nb=dv(state[f"convs.{i}.norm.bias"])))
            prev = c
        self.blocks = []
        for i, c in enumerate(cfg["dims"]):
            stage = []
            for j in range(cfg["depths"][i]):
                blk = {}
                for name, attn in (("spatial_block", "window_attn"), ("channel_block", "channel_attn")):
                    p = f"blocks.{i}.{j}.{name}."
                    d = {}
                    for cv in ("conv1", "conv2"):
                        d[cv + "_w"] = dv(state[p + cv + ".fn.dw.weight"].reshape(c, 9).t())  # tap-major [9, C]
                        d[cv + "_b"] = dv(state[p + cv + ".fn.dw.bias"])
                    d["an_w"], d["an_b"] = dv(state[p + attn + ".norm.weight"]), dv(state[p + attn + ".norm.bias"])
                    d["qkv_w"], d["qkv_b"] = dv(state[p + attn + ".fn.qkv.weight"]), dv(state[p + attn + ".fn.qkv.bias"])
                    d["proj_w"], d["proj_b"] = dv(state[p + attn + ".fn.proj.weight"]), dv(state[p + attn + ".fn.proj.bias"])
                    d["fn_w"], d["fn_b"] = dv(state[p + "ffn.norm.weight"]), dv(state[p + "ffn.norm.bias"])
                    d["fc1_w"], d["fc1_b"] = dv(state[p + "ffn.fn.net.fc1.weight"]), dv(state[p + "ffn.fn.net.fc1.bias"])
                    d["fc2_w"], d["fc2_b"] = dv(state[p + "ffn.fn.net.fc2.weight"]), dv(state[p + "ffn.fn.net.fc2.bias"])
                    blk[name] = d
                stage.append(blk)
            self.blocks.append(stage)
        self._items: Dict[Tuple[int, int], torch.Tensor] = {}

    def _window_items(self, n_windows: int, ws2: int, heads: int):
        key = (n_windows, heads)
        if key not in self._items:
            segs = [(i * ws2, (i + 1) * ws2) for i in range(n_windows)]
            self._items[key] = ops.make_items(segs, self.dev, block=ops.pick_q_block(segs, heads))
        return self._items[key]

    @staticmethod
    def _window_map(C, heads, ws, d) -> bool:
        """The un-partitioned form of the spatial block (ops.window_attention_map): head dim 32, 12 x 12 windows, a bf16 q/k/v bias to stand for the
        padded tokens.  FO1_DAVIT_WINDOW_MAP=0 keeps window_partition -> GEMM -> attention -> GEMM -> window_reverse_add (A/B, tests)."""
        return (C // heads == 32 and ws == ops.WINDOW_ATTENTION_MAP_WINDOW and d.get("qkv_b") is not None and d["qkv_b"].dtype == torch.bfloat16
                and os.environ.get("FO1_DAVIT_WINDOW_MAP", "1") != "0")

    def _window_attention(self, qkv, C, heads, ws, device):
        """softmax(q k^T / sqrt(hd)) v over the windows of ws * ws consecutive rows of the q/k/v GEMM output (modeling_davit.py:225-282).
        Head dim 32 (every DaViT stage) and windows of <= 160 tokens: one launch on the [rows, 3C] map itself; otherwise the general
        attention kernel over an item list, with V transposed into a scratch [C, rows] first."""
        n, hd = qkv.shape[0], C // heads
        if hd == 32 and ws * ws <= ops.WINDOW_ATTENTION_MAX_TOKENS:
            return ops.window_attention(qkv, C, heads, ws * ws, float(hd) ** -0.5)
        # V^T scratch [C, n padded to 64] from the owner-scoped pool (zero-initialised; the pad columns are never written):
        # this module is shared by engine replicas, so the buffer must belong to the running request, not to the module
        n_pad = _round_up(n, 64)
        vt = ops._workspace(f"davit_vt_{C}", device, C * n_pad * 2)[:C * n_pad * 2].view(torch.bfloat16).view(C, n_pad)
        ops.transpose_into(qkv[:, 2 * C:], vt, 0)
        items = self._window_items(n // (ws * ws), ws * ws, heads)
        return ops.attention(qkv[:, :C], qkv[:, C:2 * C], vt, items, heads, heads, hd, float(hd) ** -0.5, False, flops=4.0 * C * n * ws * ws)

    def _conv_ffn(self, x, H, W, d, B=1):
        """conv2 (depthwise 3x3 + residual) -> LayerNorm -> MLP(+residual); the conv and the norm are one launch."""
        x, h = ops.dwconv3x3_res_ln(x, d["conv2_w"], d["conv2_b"], H, W, d["fn_w"], d["fn_b"], 1e-5, batch=B)
        h = ops.gemm(h, d["fc1_w"], d["fc1_b"], act=ops.ACT_GELU)
        return ops.gemm(h, d["fc2_w"], d["fc2_b"], residual=x)

    def _spatial(self, x, H, W, C, heads, d, B=1):
        ws = self.cfg["window"]
        x, h = ops.dwconv3x3_res_ln(x, d["conv1_w"], d["conv1_b"], H, W, d["an_w"], d["an_b"], 1e-5, batch=B)
        if self._window_map(C, heads, ws, d):
            # no partition, no padded rows in the GEMMs, no reverse: the attention finds a window's tokens among the pixel rows by arithmetic, the
            # tokens of the reference's zero padding (:248-251) are the q/k/v bias row, and the residual rides in the proj GEMM's epilogue
            qkv = ops.gemm(h, d["qkv_w"], d["qkv_b"])
            att = ops.window_attention_map(qkv, C, heads, ws, H, W, B, d["qkv_b"], float(C // heads) ** -0.5)
            x = ops.gemm(att, d["proj_w"], d["proj_b"], residual=x)
            return self._conv_ffn(x, H, W, d, B)
        hw = ops.window_partition(h, H, W, ws, batch=B)   # zero-padded AFTER the norm, like the reference (:248-251)
        qkv = ops.gemm(hw, d["qkv_w"], d["qkv_b"])
        att = self._window_attention(qkv, C, heads, ws, x.device)
        y = ops.gemm(att, d["proj_w"], d["proj_b"])
        x = ops.window_reverse_add(y, x, H, W, ws, batch=B)
        return self._conv_ffn(x, H, W, d, B)

    def _channel(self, x, H, W, C, d, B=1):
        x, h = ops.dwconv3x3_res_ln(x, d["conv1_w"], d["conv1_b"], H, W, d["an_w"], d["an_b"], 1e-5, batch=B)
        qkv = ops.gemm(h, d["qkv_w"], d["qkv_b"])
        a = ops.channel_attention(qkv, C, batch=B)
        x = ops.gemm(a, d["proj_w"], d["proj_b"], residual=x)
        return self._conv_ffn(x, H, W, d, B)

    # ---- ragged batches: images of different sizes in ONE pass (row-wise packing, per-image geometry tables) ------------------------
    def ragged_plan(self, sizes: Sequence[Tuple[int, int]]) -> "RaggedAuxPlan":
        key = tuple((int(h), int(w)) for h, w in sizes)
        plans = self.__dict__.setdefault("_rplans", {})
        pl = plans.get(key)
        if pl is None:
            if len(plans) >= 64:
                plans.pop(next(iter(plans)))
            pl = plans[key] = RaggedAuxPlan(key, self.cfg, self.dev)
        return pl

    def _conv_ffn_var(self, x, pix, d):
        x, h = ops.dwconv3x3_res_ln_var(x, d["conv2_w"], d["conv2_b"], pix, d["fn_w"], d["fn_b"], 1e-5)
        h = ops.gemm(h, d["fc1_w"], d["fc1_b"], act=ops.ACT_GELU)
        return ops.gemm(h, d["fc2_w"], d["fc2_b"], residual=x)

    def _spatial_var(self, x, lv, C, heads, d):
        ws = self.cfg["window"]
        x, h = ops.dwconv3x3_res_ln_var(x, d["conv1_w"], d["conv1_b"], lv["pix"], d["an_w"], d["an_b"], 1e-5)
        if self._window_map(C, heads, ws, d):
            qkv = ops.gemm(h, d["qkv_w"], d["qkv_b"])
            att = ops.window_attention_map_var(qkv, C, heads, ws, lv["win"], d["qkv_b"], float(C // heads) ** -0.5)
            x = ops.gemm(att, d["proj_w"], d["proj_b"], residual=x)
            return self._conv_ffn_var(x, lv["pix"], d)
        hw = ops.window_partition_var(h, lv["win"], ws)         # per image zero-padded AFTER the norm, like the reference (:248-251)
        qkv = ops.gemm(hw, d["qkv_w"], d["qkv_b"])
        att = self._window_attention(qkv, C, heads, ws, x.device)
        y = ops.gemm(att, d["proj_w"], d["proj_b"])
        x = ops.window_reverse_add_var(y, x, lv["win"], ws)
        return self._conv_ffn_var(x, lv["pix"], d)

    def _channel_var(self, x, lv, C, d):
        x, h = ops.dwconv3x3_res_ln_var(x, d["conv1_w"], d["conv1_b"], lv["pix"], d["an_w"], d["an_b"], 1e-5)
        qkv = ops.gemm(h, d["qkv_w"], d["qkv_b"])
        a = ops.channel_attention_var(qkv, C, lv["tok"])
        x = ops.gemm(a, d["proj_w"], d["proj_b"], residual=x)
        return self._conv_ffn_var(x, lv["pix"], d)

    def forward_ragged(self, imgs: Sequence[torch.Tensor]):
        """imgs: [3,H_b,W_b] device tensors of DIFFERENT sizes -> ([4 token-major maps [sum_b H_ib*W_ib, C_i] bf16, image b at rows
        plan.row0[i][b] ...], plan (RaggedAuxPlan: per level sizes[i][b], row0[i][b])).  Same launches as forward(), every GEMM /
        LayerNorm over the rows of ALL images, the spatial kernels per image through the geometry tables: each image's maps are
        bit-identical to its one-image pass when the GEMM tile is pinned (tests/test_ragged_towers_gpu.py).  The reference runs the
        tower image by image (davit_aux_encoder.py:54-69)."""
        cfg = self.cfg
        plan = self.ragged_plan([tuple(im.shape[-2:]) for im in imgs])
        x = torch.empty(plan.total_in, 8, dtype=torch.bfloat16, device=self.dev)
        for im, r0, (H, W) in zip(imgs, plan.in_row0, plan.in_sizes):
            ops.nchw_to_hwc8(im.contiguous(), out=x[r0:r0 + H * W])
        outs = []
        for i, C in enumerate(cfg["dims"]):
            cv, lv = self.convs[i], plan.levels[i]
            k, s, p = cfg["patch_size"][i], cfg["patch_stride"][i], cfg["patch_padding"][i]
            prev_sizes = plan.in_sizes if i == 0 else plan.sizes[i - 1]
            if i > 0 and cfg["patch_prenorm"][i] and cv["Kp"] == k * k * x.shape[1] and ops.conv3x3_implicit_for(prev_sizes, s, C, x.shape[1], k, p):
                # pre-norm ConvEmbed as an implicit GEMM over the ragged pack (as forward() does for uniform batches): one zero-framed buffer with a
                # common row pitch, the same bits as layernorm + im2col_var + gemm without the [M, 9 Cin] column matrix
                cp = ops.conv3x3_plan(prev_sizes, s, x.shape[1], self.dev)
                assert cp.out_hw == [tuple(t) for t in plan.sizes[i]] and cp.M_in == x.shape[0]
                xp = ops.layernorm_rows(x, cv["nw"], cv["nb"], 1e-5, torch.zeros(cp.pad_rows, x.shape[1], dtype=torch.bfloat16, device=self.dev), cp.rowmap)
                x = ops.conv3x3_gemm(xp, cp, cv["w"], cv["b"])
            else:
                if i > 0 and cfg["patch_prenorm"][i]:
                    x = ops.layernorm(x, cv["nw"], cv["nb"], 1e-5)
                col = ops.im2col_var(x, lv["conv"], k, k, s, p, ld=cv["Kp"])
                x = ops.gemm(col, cv["w"], cv["b"])
                if i == 0 or not cfg["patch_prenorm"][i]:
                    x = ops.layernorm(x, cv["nw"], cv["nb"], 1e-5)
            for blk in self.blocks[i]:
                x = self._spatial_var(x, lv, C, cfg["heads"][i], blk["spatial_block"])
                x = self._channel_var(x, lv, C, blk["channel_block"])
            outs.append(x)
        return outs, plan

    def forward(self, img: torch.Tensor):
        """img [3,H,W] or [B,3,H,W] (device, bf16/fp32, CLIP-normalised; B same-size images in one pass).  Returns
        ([4 token-major maps [B*H_i*W_i, C_i] bf16 — image b at rows [b*H_i*W_i, (b+1)*H_i*W_i)], [(H_i, W_i)])."""
        cfg = self.cfg
        from . import stage_abi
        if stage_abi.enabled():      # the same launches, sequenced by fo1_davit_forward (csrc/stages.hip)
            return stage_abi.davit_stage(self).forward(img)
        if img.dim() == 3:
            img = img.unsqueeze(0)
        B = img.shape[0]
        H, W = img.shape[2:]
        x = ops.nchw_to_hwc8(img.contiguous())
        outs, sizes = [], []
        for i, C in enumerate(cfg["dims"]):
            cv = self.convs[i]
            k, s, p = cfg["patch_size"][i], cfg["patch_stride"][i], cfg["patch_padding"][i]
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            if i > 0 and cfg["patch_prenorm"][i] and cv["Kp"] == k * k * x.shape[1] and ops.conv3x3_implicit_for(((H, W),) * B, s, C, x.shape[1], k, p):
                # pre-norm ConvEmbed (modeling_davit.py:102-148) as an implicit GEMM: the LayerNorm writes the zero-padded map, the 256 x 256 GEMM
                # gathers its 9 taps from it — no [M, 9 Cin] column matrix (same bits as layernorm + im2col + gemm)
                pl = ops.conv3x3_plan(((H, W),) * B, s, x.shape[1], self.dev)
                xp = ops.layernorm_rows(x, cv["nw"], cv["nb"], 1e-5, ops.conv3x3_padded(pl, self.dev), pl.rowmap)
                x = ops.conv3x3_gemm(xp, pl, cv["w"], cv["b"])
                H, W = Ho, Wo
            else:
                if i > 0 and cfg["patch_prenorm"][i]:
                    x = ops.layernorm(x, cv["nw"], cv["nb"], 1e-5)
                col, H, W = ops.im2col(x, H, W, k, k, s, p, ld=cv["Kp"], batch=B)
                x = ops.gemm(col, cv["w"], cv["b"])
            if i == 0 or not cfg["patch_prenorm"][i]:
                x = ops.layernorm(x, cv["nw"], cv["nb"], 1e-5)
            for blk in self.blocks[i]:
                x = self._spatial(x, H, W, C, cfg["heads"][i], blk["spatial_block"], B)
                x = self._channel(x, H, W, C, blk["channel_block"], B)
            outs.append(x)
            sizes.append((H, W))
        return outs, sizes


class RaggedAuxPlan:
    """Geometry tables (ops.ImgSegs) of a ragged DaViT pass over images of sizes [(H_b, W_b)]: per stage i the conv-embed table
    (previous level -> this level), the pixel table, the window table (ws = 12: windows down / across, first window row) and the
    token table of the channel attention.  sizes[i][b] = (H_ib, W_ib), row0[i][b] = first row of image b in the level-i map."""

    def __init__(self, sizes, cfg, device):
        self.in_sizes = list(sizes)
        self.in_row0, off = [], 0
        for H, W in sizes:
            self.in_row0.append(off)
            off += H * W
        self.total_in = off
        ws = cfg["window"]
        self.sizes, self.row0, self.levels = [], [], []
        prev_sizes, prev_row0 = self.in_sizes, self.in_row0
        for i in range(len(cfg["dims"])):
            k, s, p = cfg["patch_size"][i], cfg["patch_stride"][i], cfg["patch_padding"][i]
            cur = [((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1) for H, W in prev_sizes]
            r0, off = [], 0
            for h, w in cur:
                r0.append(off)
                off += h * w
            npx = [h * w for h, w in cur]
            nwin = [(-(-h // ws), -(-w // ws)) for h, w in cur]
            wr0, woff = [], 0
            for a, b in nwin:
                wr0.append(woff)
                woff += a * b * ws * ws
            pin = [h * w for h, w in prev_sizes]
            conv = ops.ImgSegs([(pr, ph, pw, r, h, w) for pr, (ph, pw), r, (h, w) in zip(prev_row0, prev_sizes, r0, cur)], device,
                               max(pin), sum(pin), max(npx), sum(npx))
            pix = ops.ImgSegs([(r, h, w) for r, (h, w) in zip(r0, cur)], device, max(npx), sum(npx), max(npx), sum(npx))
            win = ops.ImgSegs([(r, h, w, wr, a, b) for r, (h, w), wr, (a, b) in zip(r0, cur, wr0, nwin)], device, max(npx), sum(npx),
                              max(a * b * ws * ws for a, b in nwin), woff)
            tok = ops.ImgSegs([(r, n) for r, n in zip(r0, npx)], device, max(npx), sum(npx), max(npx), sum(npx))
            self.levels.append(dict(conv=conv, pix=pix, win=win, tok=tok))
            self.sizes.append(cur)
            self.row0.append(r0)
            prev_sizes, prev_row0 = cur, r0
