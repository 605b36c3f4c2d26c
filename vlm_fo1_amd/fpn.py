"""ViTDet SimpleFPN on the MI355X engine.

Mirrors `SimpleFP.forward` (simple_fpn.py:100-216) as instantiated by the HFRE
(hybrid_finegrained_region_encoder.py:175, strides [3.5, 7, 14, 28] at :245) on the token-major
last ViT map.  ConvTranspose2d(k=2,s=2) = GEMM [n, Cin] x [4*Cout, Cin]^T + pixel shuffle;
1x1 conv = GEMM; 3x3 conv = im2col + GEMM; the channel LayerNorm (eps 1e-6, biased variance) is the
row LayerNorm on token-major maps; MaxPool2d(2,2) is its own kernel."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import ops


class SimpleFPN:
    STAGES = ("simfp_1", "simfp_2", "simfp_3", "simfp_4")

    def __init__(self, state: Dict[str, torch.Tensor], device):
        self.dev = torch.device(device)
        bf = torch.bfloat16

        def dv(t):
            return t.to(device=self.dev, dtype=bf).contiguous()

        def convT(w, b):  # [Cin, Cout, 2, 2] -> GEMM weight rows (dy, dx, co), bias tiled x4
            cin, cout = w.shape[:2]
            return dv(w.permute(2, 3, 1, 0).reshape(4 * cout, cin)), dv(b.repeat(4)), cout

        def conv1(w):
            return dv(w.reshape(w.shape[0], w.shape[1]))

        def conv3(w):  # [Cout, Cin, 3, 3] -> [Cout, (ky,kx,cin)]
            return dv(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))

        s = state
        self.t1a = convT(s["simfp_1.0.weight"], s["simfp_1.0.bias"])
        self.t1_ln = (dv(s["simfp_1.1.weight"]), dv(s["simfp_1.1.bias"]))
        self.t1b = convT(s["simfp_1.3.weight"], s["simfp_1.3.bias"])
        self.t2 = convT(s["simfp_2.0.weight"], s["simfp_2.0.bias"])
        self.heads = []
        for name, a, b in (("simfp_1", "4.", "5."), ("simfp_2", "1.", "2."), ("simfp_3", "0.", "1."), ("simfp_4", "1.", "2.")):
            self.heads.append(dict(
                w1=conv1(s[f"{name}.{a}weight"]), n1=(dv(s[f"{name}.{a}norm.weight"]), dv(s[f"{name}.{a}norm.bias"])),
                w3=conv3(s[f"{name}.{b}weight"]), n3=(dv(s[f"{name}.{b}norm.weight"]), dv(s[f"{name}.{b}norm.bias"]))))

    def _up(self, x, H, W, t, B=1):
        w, b, cout = t
        return ops.pixel_shuffle2(ops.gemm(x, w, b), H, W, cout, batch=B), 2 * H, 2 * W

    def _head(self, x, H, W, h, B=1):
        y = ops.gemm(x, h["w1"])
        C = y.shape[1]
        if ops.conv3x3_implicit_for(((H, W),) * B, 1, h["w3"].shape[0], C, 3, 1):
            # the 3x3 output convolution (simple_fpn.py:141-176) as an implicit GEMM: the channel LayerNorm writes the zero-padded map, the GEMM
            # gathers its taps from it (no im2col matrix: 2.2 GB per launch at the finest level); same bits as layernorm + im2col + gemm
            pl = ops.conv3x3_plan(((H, W),) * B, 1, C, y.device)
            yp = ops.layernorm_rows(y, h["n1"][0], h["n1"][1], 1e-6, ops.conv3x3_padded(pl, y.device), pl.rowmap)
            y = ops.conv3x3_gemm(yp, pl, h["w3"])
        else:
            y = ops.layernorm(y, h["n1"][0], h["n1"][1], 1e-6)
            col, _, _ = ops.im2col(y, H, W, 3, 3, 1, 1, batch=B)
            y = ops.gemm(col, h["w3"])
        return ops.layernorm(y, h["n3"][0], h["n3"][1], 1e-6)

    # ---- ragged batches: maps of different grids packed row-wise ----------------------------------------------------------------
    def ragged_plan(self, grids, row0) -> "RaggedFpnPlan":
        key = (tuple((int(a), int(b)) for a, b in grids), tuple(int(r) for r in row0))
        plans = self.__dict__.setdefault("_rplans", {})
        pl = plans.get(key)
        if pl is None:
            if len(plans) >= 64:
                plans.pop(next(iter(plans)))
            pl = plans[key] = RaggedFpnPlan(key[0], key[1], self.dev)
        return pl

    def _head_var(self, x, sg, h, sizes):
        y = ops.gemm(x, h["w1"])
        C = y.shape[1]
        if ops.conv3x3_implicit_for(sizes, 1, h["w3"].shape[0], C, 3, 1):
            # the 3x3 output convolution as an implicit GEMM over the ragged pack (levels are packed back to back, image by image: RaggedFpnPlan)
            cp = ops.conv3x3_plan(sizes, 1, C, y.device)
            assert cp.M_in == y.shape[0]
            yp = ops.layernorm_rows(y, h["n1"][0], h["n1"][1], 1e-6, torch.zeros(cp.pad_rows, C, dtype=torch.bfloat16, device=y.device), cp.rowmap)
            y = ops.conv3x3_gemm(yp, cp, h["w3"])
        else:
            y = ops.layernorm(y, h["n1"][0], h["n1"][1], 1e-6)
            col = ops.im2col_var(y, sg, 3, 3, 1, 1)
            y = ops.gemm(col, h["w3"])
        return ops.layernorm(y, h["n3"][0], h["n3"][1], 1e-6)

    def forward_ragged(self, x: torch.Tensor, grids, row0):
        """x [sum gh_b*gw_b, 1280] token-major raster maps of images with DIFFERENT grids (image b at rows row0[b]...) -> (4 maps
        [sum .., 512], plan with sizes[l][b] / row0[l][b]).  Same launches as forward(); every image bit-identical to its own pass."""
        pl = self.ragged_plan(grids, row0)

        def up(y, sg, t):
            w, b, cout = t
            return ops.pixel_shuffle2_var(ops.gemm(y, w, b), sg, cout)

        outs = []
        y = up(x, pl.up_1a, self.t1a)
        y = ops.layernorm(y, self.t1_ln[0], self.t1_ln[1], 1e-6)
        y = ops.bias_act(y, None, 1)
        y = up(y, pl.up_1b, self.t1b)
        outs.append(self._head_var(y, pl.conv[0], self.heads[0], pl.sizes[0]))
        y = up(x, pl.up_1a, self.t2)
        outs.append(self._head_var(y, pl.conv[1], self.heads[1], pl.sizes[1]))
        outs.append(self._head_var(x, pl.conv[2], self.heads[2], pl.sizes[2]))
        y = ops.maxpool2_var(x, pl.pool)
        outs.append(self._head_var(y, pl.conv[3], self.heads[3], pl.sizes[3]))
        return outs, pl

    def forward(self, x: torch.Tensor, H: int, W: int, batch: int = 1) -> Tuple[List[torch.Tensor], List[Tuple[int, int]]]:
        """x [batch*H*W, 1280] token-major bf16 (same-size maps stacked) -> 4 token-major maps [batch*.., 512] at (4H,4W),
        (2H,2W), (H,W), (H/2,W/2); image b owns rows [b*h*w, (b+1)*h*w) of every level."""
        B = batch
        from . import stage_abi
        if stage_abi.enabled():      # the same launches, sequenced by fo1_simplefpn_forward (csrc/stages.hip)
            return stage_abi.fpn_stage(self).forward(x, H, W, B)
        outs, sizes = [], []
        y, h1, w1 = self._up(x, H, W, self.t1a, B)
        y = ops.layernorm(y, self.t1_ln[0], self.t1_ln[1], 1e-6)
        y = ops.bias_act(y, None, 1)
        y, h1, w1 = self._up(y, h1, w1, self.t1b, B)
        outs.append(self._head(y, h1, w1, self.heads[0], B)); sizes.append((h1, w1))
        y, h2, w2 = self._up(x, H, W, self.t2, B)
        outs.append(self._head(y, h2, w2, self.heads[1], B)); sizes.append((h2, w2))
        outs.append(self._head(x, H, W, self.heads[2], B)); sizes.append((H, W))
        y = ops.maxpool2(x, H, W, batch=B)
        outs.append(self._head(y, H // 2, W // 2, self.heads[3], B)); sizes.append((H // 2, W // 2))
        return outs, sizes


class RaggedFpnPlan:
    """Geometry tables of a ragged SimpleFPN pass: grids[b] = (gh, gw) of image b whose raster map starts at row0[b] of the input.
    sizes[l][b] / row0[l][b] describe output level l (strides 3.5, 7, 14, 28: 4x, 2x, 1x, 1/2 of the grid)."""

    def __init__(self, grids, row0, device):
        B = len(grids)

        def pack(sz):
            r, off = [], 0
            for h, w in sz:
                r.append(off)
                off += h * w
            return r, off

        g1 = list(grids)
        g2 = [(2 * h, 2 * w) for h, w in g1]
        g4 = [(4 * h, 4 * w) for h, w in g1]
        gh = [(h // 2, w // 2) for h, w in g1]
        r1 = list(row0)
        n1 = [h * w for h, w in g1]
        self.contiguous_input = all(r1[i] + n1[i] == (r1[i + 1] if i + 1 < B else r1[i] + n1[i]) for i in range(B)) and r1[0] == 0
        if not self.contiguous_input:
            raise ValueError("SimpleFPN.forward_ragged: the images' rows must be packed back to back (vit.BatchPlan order)")
        r2, t2 = pack(g2)
        r4, t4 = pack(g4)
        rh, th = pack(gh)
        n2, n4, nh = [h * w for h, w in g2], [h * w for h, w in g4], [h * w for h, w in gh]
        S = ops.ImgSegs
        self.up_1a = S([(a, h, w, b) for a, (h, w), b in zip(r1, g1, r2)], device, max(n1), sum(n1), max(n2), t2)        # grid -> 2x
        self.up_1b = S([(a, h, w, b) for a, (h, w), b in zip(r2, g2, r4)], device, max(n2), t2, max(n4), t4)             # 2x -> 4x
        self.pool = S([(a, h, w, b, h // 2, w // 2) for a, (h, w), b in zip(r1, g1, rh)], device, max(n1), sum(n1), max(nh), th)

        def same(rr, gg, nn, tot):      # 3x3 / stride 1 / pad 1: output geometry = input geometry
            return S([(a, h, w, a, h, w) for a, (h, w) in zip(rr, gg)], device, max(nn), tot, max(nn), tot)

        self.conv = [same(r4, g4, n4, t4), same(r2, g2, n2, t2), same(r1, g1, n1, sum(n1)), same(rh, gh, nh, th)]
        self.sizes = [g4, g2, g1, gh]
        self.row0 = [r4, r2, r1, rh]
