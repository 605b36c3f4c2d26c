"""UPN proposal detector on MI355X — the deformable-transformer stages, host orchestration over libfo1hip.so.

Reference: detect_tools/upn/models/encoder/upn_encoder.py (DeformableTransformerEncoderLayer :62-110, UPNEncoder :198-288) and
detect_tools/upn/ops/modules/ms_deform_attn.py (MSDeformAttn.forward :100-204).  One image per call, token-major bf16 rows
[S, 256] (S = sum of the pyramid's H_l W_l, level-major); no padding mask (a single image has none).

MI355X-first shape of a layer (5 launches for the attention half instead of the reference's ~12 torch ops):
    q      = add(src, pos)                                    fo1_add_bf16
    value  = GEMM(src, W_value) + b                           bf16
    ol     = GEMM(q, [W_offsets ; W_attention_weights]) + b   ONE GEMM, fp32 out: [S, 480] = offsets 320 | logits 160
    attn   = msda_fused(value, ol, reference points)          softmax + sampling locations + bilinear gather in one kernel:
                                                              the [S,8,5,4,2] locations and [S,8,5,4] weights never exist
    src    = LayerNorm(GEMM(attn, W_out) + b + src)           residual in the GEMM epilogue
    src    = LayerNorm(GEMM(ReLU(GEMM(src, W1) + b1), W2) + b2 + src)
No eager / CPU fallback: every tensor op is a kernel of the library; torch only owns the memory."""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

BF = torch.bfloat16


def _dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=BF).contiguous()


class MSDeformAttnWeights:
    """Device weights of one MSDeformAttn: value_proj, [sampling_offsets ; attention_weights] stacked into one GEMM, output_proj."""

    def __init__(self, state: Dict[str, torch.Tensor], prefix: str, device, n_heads: int, n_levels: int, n_points: int):
        self.M, self.L, self.P = n_heads, n_levels, n_points
        n_off, n_aw = n_heads * n_levels * n_points * 2, n_heads * n_levels * n_points
        so_w, so_b = state[prefix + "sampling_offsets.weight"], state[prefix + "sampling_offsets.bias"]
        aw_w, aw_b = state[prefix + "attention_weights.weight"], state[prefix + "attention_weights.bias"]
        if tuple(so_w.shape) != (n_off, so_w.shape[1]) or aw_w.shape[0] != n_aw:
            raise ValueError(f"{prefix}: sampling_offsets / attention_weights do not match heads x levels x points = {n_heads} x {n_levels} x {n_points}")
        self.w_ol = _dev(torch.cat([so_w, aw_w], 0), device)
        self.b_ol = _dev(torch.cat([so_b, aw_b], 0), device)
        self.w_v, self.b_v = _dev(state[prefix + "value_proj.weight"], device), _dev(state[prefix + "value_proj.bias"], device)
        self.w_o, self.b_o = _dev(state[prefix + "output_proj.weight"], device), _dev(state[prefix + "output_proj.bias"], device)

    def attend(self, query: torch.Tensor, ref: torch.Tensor, value_in: torch.Tensor, shapes_dev: torch.Tensor, start_dev: torch.Tensor,
               residual: torch.Tensor) -> torch.Tensor:
        """query [Lq, C] (position embedding already added), ref fp32 [1, Lq, L, 2|4], value_in [S, C] -> output_proj(...) + residual."""
        value = ops.gemm(value_in, self.w_v, self.b_v)
        ol = ops.gemm(query, self.w_ol, self.b_ol, out_f32=True)
        att = ops.msda_fused(value.unsqueeze(0), shapes_dev, start_dev, ol.unsqueeze(0), ref, self.M, self.P)
        return ops.gemm(att[0], self.w_o, self.b_o, residual=residual)


def encoder_reference_points(shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """UPNEncoder.get_reference_points (upn_encoder.py:198-213) for one unpadded image (valid ratios 1): fp32 [1, S, L, 2], host."""
    pts = []
    for H, W in shapes:
        ys = (np.arange(H, dtype=np.float32) + 0.5) / np.float32(H)
        xs = (np.arange(W, dtype=np.float32) + 0.5) / np.float32(W)
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        pts.append(np.stack([xx.reshape(-1), yy.reshape(-1)], -1))
    ref = np.concatenate(pts, 0)                                    # [S, 2]
    return torch.from_numpy(np.repeat(ref[None, :, None, :], len(shapes), 2).copy())


class DeformableEncoder:
    """UPNEncoder (6 x DeformableTransformerEncoderLayer in configs/upn_large.py) for one image."""

    def __init__(self, state: Dict[str, torch.Tensor], prefix: str, n_layers: int, device="cuda", n_heads: int = 8, n_levels: int = 5,
                 n_points: int = 4):
        self.dev = torch.device(device)
        self.layers = []
        for i in range(n_layers):
            p = f"{prefix}layers.{i}."
            self.layers.append(dict(
                attn=MSDeformAttnWeights(state, p + "self_attn.", self.dev, n_heads, n_levels, n_points),
                n1w=_dev(state[p + "norm1.weight"], self.dev), n1b=_dev(state[p + "norm1.bias"], self.dev),
                w1=_dev(state[p + "linear1.weight"], self.dev), b1=_dev(state[p + "linear1.bias"], self.dev),
                w2=_dev(state[p + "linear2.weight"], self.dev), b2=_dev(state[p + "linear2.bias"], self.dev),
                n2w=_dev(state[p + "norm2.weight"], self.dev), n2b=_dev(state[p + "norm2.bias"], self.dev)))
        self.n_levels = n_levels
        self._plans: Dict[tuple, tuple] = {}

    def plan(self, shapes: Sequence[Tuple[int, int]]):
        key = tuple(shapes)
        if key not in self._plans:
            if len(shapes) != self.n_levels:
                raise ValueError(f"encoder built for {self.n_levels} levels, got {len(shapes)}")
            start = [0]
            for h, w in shapes[:-1]:
                start.append(start[-1] + h * w)
            self._plans[key] = (torch.tensor(shapes, dtype=torch.int64, device=self.dev), torch.tensor(start, dtype=torch.int64, device=self.dev),
                                encoder_reference_points(shapes).to(self.dev))
        return self._plans[key]

    def forward(self, src: torch.Tensor, pos: torch.Tensor, shapes: Sequence[Tuple[int, int]], collect: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """src, pos: bf16 [S, C] on the device (pos = position + level embedding) -> memory bf16 [S, C]."""
        if src.dtype != BF or pos.dtype != BF or not src.is_cuda:
            raise TypeError("DeformableEncoder: src / pos must be bf16 GPU tensors (no CPU path exists)")
        shapes_dev, start_dev, ref = self.plan(shapes)
        if src.shape[0] != sum(h * w for h, w in shapes):
            raise ValueError("DeformableEncoder: src rows do not match the pyramid")
        x = src
        for ly in self.layers:
            q = ops.add(x, pos)
            y = ly["attn"].attend(q, ref, x, shapes_dev, start_dev, residual=x)
            x = ops.layernorm(y, ly["n1w"], ly["n1b"], 1e-5)
            h = ops.gemm(x, ly["w1"], ly["b1"], act=ops.ACT_RELU)
            y = ops.gemm(h, ly["w2"], ly["b2"], residual=x)
            x = ops.layernorm(y, ly["n2w"], ly["n2b"], 1e-5)
            if collect is not None:
                collect.append(x)
        return x
