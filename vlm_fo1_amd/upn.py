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


# ---- query selection, decoder, prediction heads ---------------------------------------------------------------------------------
# Reference: models/architecture/deformable_transformer.py:262-336 (get_two_stage_proposal), models/utils/detr_utils.py:351-415
# (gen_encoder_output_proposals), models/decoder/upn_decoder.py:98-139 (layer), :262-378 (UPNDecoder.forward),
# models/architecture/upn_model.py:96-140 (box / class heads; ContrastiveAssign = dot product with the prompt embedding).
# Box coordinates stay fp32 end to end (logit-space arithmetic is precision-sensitive near 0 and 1); features are bf16.
def _pad_rows(w: torch.Tensor, b: Optional[torch.Tensor], rows: int):
    """zero-pad a small head ([4, C] boxes, [1, C] prompt) to `rows` output features so the fp32 GEMM output rows are 32-byte aligned"""
    wp = torch.zeros(rows, w.shape[1], dtype=w.dtype)
    wp[:w.shape[0]] = w
    bp = None
    if b is not None:
        bp = torch.zeros(rows, dtype=b.dtype)
        bp[:b.shape[0]] = b
    return wp, bp


class _MLP:
    """models/module/mlp.py: Linear + ReLU ... Linear; the last layer can be emitted in fp32 (box deltas)."""

    def __init__(self, state, prefix, n, device, pad_last: int = 0):
        self.layers = []
        for i in range(n):
            w, b = state[f"{prefix}layers.{i}.weight"], state[f"{prefix}layers.{i}.bias"]
            if i == n - 1 and pad_last:
                w, b = _pad_rows(w, b, pad_last)
            self.layers.append((_dev(w, device), _dev(b, device)))

    def __call__(self, x, last_f32=False):
        n = len(self.layers)
        for i, (w, b) in enumerate(self.layers):
            if i < n - 1:
                x = ops.gemm(x, w, b, act=ops.ACT_RELU)
            else:
                x = ops.gemm(x, w, b, out_f32=last_f32)
        return x


def encoder_output_proposals(shapes: Sequence[Tuple[int, int]]):
    """gen_encoder_output_proposals for one unpadded image, on the host: (keep uint8 [S], proposals in logit space fp32 [S, 4])."""
    props = []
    for lvl, (H, W) in enumerate(shapes):
        gy, gx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        grid = (np.stack([gx, gy], -1) + np.float32(0.5)) / np.array([W, H], dtype=np.float32)
        wh = np.ones_like(grid) * np.float32(0.05) * np.float32(2.0 ** lvl)
        props.append(np.concatenate([grid, wh], -1).reshape(-1, 4))
    p = torch.from_numpy(np.concatenate(props, 0).astype(np.float32))
    valid = ((p > 0.01) & (p < 0.99)).all(-1)
    logit = torch.log(p / (1 - p)).masked_fill(~valid[:, None], float("inf"))
    return valid.to(torch.uint8), logit


class QuerySelector:
    """Two-stage proposal generation: per-token score (contrast with the universal prompt) and box, top-k tokens -> decoder reference boxes."""

    def __init__(self, state, device="cuda", n_queries: int = 900, prefix: str = "transformer."):
        self.dev = torch.device(device)
        self.nq = n_queries
        self.w_enc, self.b_enc = _dev(state[prefix + "enc_output.weight"], self.dev), _dev(state[prefix + "enc_output.bias"], self.dev)
        self.nw, self.nb = _dev(state[prefix + "enc_output_norm.weight"], self.dev), _dev(state[prefix + "enc_output_norm.bias"], self.dev)
        self.box = _MLP(state, prefix + "enc_out_bbox_embed.", 3, self.dev, pad_last=8)
        self.prompts = {k: _dev(_pad_rows(state[f"{prefix}{k}.weight"], None, 8)[0], self.dev) for k in ("fine_grained_prompt", "coarse_grained_prompt")}
        self._plans = {}

    def plan(self, shapes):
        key = tuple(shapes)
        if key not in self._plans:
            keep, props = encoder_output_proposals(shapes)
            self._plans[key] = (keep.to(self.dev), props.to(self.dev).contiguous())
        return self._plans[key]

    def forward(self, memory: torch.Tensor, shapes, prompt_type: str = "fine_grained_prompt") -> dict:
        keep, props = self.plan(shapes)
        S = memory.shape[0]
        om = ops.mask_rows(memory, keep)
        om = ops.layernorm(ops.gemm(om, self.w_enc, self.b_enc), self.nw, self.nb, 1e-5)
        sc = ops.gemm(om, self.prompts[prompt_type], out_f32=True)                 # [S, 8] fp32, column 0 = the score
        coords = ops.box_refine(self.box(om, last_f32=True), props, mode=1)        # logit space; +inf rows for invalid proposals
        idx, val = ops.topk_desc(sc, min(self.nq, S), stride=8, n=S)
        return dict(scores=sc, coords=coords, idx=idx, topk_scores=val, refpoints=ops.gather_rows_f32(coords, idx))


class DeformableDecoder:
    """UPNDecoder (6 x DeformableTransformerDecoderLayer in configs/upn_large.py) + the box / class heads for one image."""

    def __init__(self, state, n_layers: int, device="cuda", n_queries: int = 900, n_heads: int = 8, n_levels: int = 5, n_points: int = 4,
                 prefix: str = "transformer."):
        self.dev = torch.device(device)
        self.nq, self.H, self.L = n_queries, n_heads, n_levels
        d = prefix + "decoder."
        self.layers = []
        for i in range(n_layers):
            p = f"{d}layers.{i}."
            C = state[p + "self_attn.out_proj.weight"].shape[0]
            wi, bi = state[p + "self_attn.in_proj_weight"], state[p + "self_attn.in_proj_bias"]
            self.layers.append(dict(
                cross=MSDeformAttnWeights(state, p + "cross_attn.", self.dev, n_heads, n_levels, n_points),
                w_qk=_dev(wi[:2 * C], self.dev), b_qk=_dev(bi[:2 * C], self.dev), w_v=_dev(wi[2 * C:], self.dev), b_v=_dev(bi[2 * C:], self.dev),
                w_o=_dev(state[p + "self_attn.out_proj.weight"], self.dev), b_o=_dev(state[p + "self_attn.out_proj.bias"], self.dev),
                n1=(_dev(state[p + "norm1.weight"], self.dev), _dev(state[p + "norm1.bias"], self.dev)),
                n2=(_dev(state[p + "norm2.weight"], self.dev), _dev(state[p + "norm2.bias"], self.dev)),
                n3=(_dev(state[p + "norm3.weight"], self.dev), _dev(state[p + "norm3.bias"], self.dev)),
                w1=_dev(state[p + "linear1.weight"], self.dev), b1=_dev(state[p + "linear1.bias"], self.dev),
                w2=_dev(state[p + "linear2.weight"], self.dev), b2=_dev(state[p + "linear2.bias"], self.dev)))
        self.C = C
        self.norm = (_dev(state[d + "norm.weight"], self.dev), _dev(state[d + "norm.bias"], self.dev))
        self.ref_head = _MLP(state, d + "ref_point_head.", 2, self.dev)
        self.bbox = _MLP(state, "bbox_embed.0.", 3, self.dev, pad_last=8)          # shared by the layers and the final head (dec_pred_bbox_embed_share)
        self.tgt = _dev(state[prefix + "tgt_embed.weight"], self.dev)
        self.prompts = {k: _dev(_pad_rows(state[f"{prefix}{k}.weight"], None, 8)[0], self.dev) for k in ("fine_grained_prompt", "coarse_grained_prompt")}
        self.items = ops.make_items([(0, n_queries)], self.dev, block=64)
        self.zero8 = torch.zeros(n_queries, 8, dtype=torch.float32, device=self.dev)
        self._plans = {}

    def plan(self, shapes):
        key = tuple(shapes)
        if key not in self._plans:
            start = [0]
            for h, w in shapes[:-1]:
                start.append(start[-1] + h * w)
            self._plans[key] = (torch.tensor(shapes, dtype=torch.int64, device=self.dev), torch.tensor(start, dtype=torch.int64, device=self.dev))
        return self._plans[key]

    def forward(self, memory: torch.Tensor, shapes, refpoints_unsig: torch.Tensor, prompt_type: str = "fine_grained_prompt") -> dict:
        """memory bf16 [S, C]; refpoints_unsig fp32 [nq, 4] (logit space) -> hs (decoder.norm applied) per layer, reference boxes per layer,
        pred_boxes fp32 [nq, 4] (cx, cy, w, h in [0, 1]) and pred_logits fp32 [nq]."""
        nq, C, H = self.nq, self.C, self.H
        if tuple(refpoints_unsig.shape) != (nq, 4) or refpoints_unsig.dtype != torch.float32:
            raise ValueError("DeformableDecoder: refpoints must be fp32 [n_queries, 4]")
        shapes_dev, start_dev = self.plan(shapes)
        ref = ops.box_refine(self.zero8, refpoints_unsig.contiguous(), mode=2)      # sigmoid
        refs, hs = [ref], []
        out = self.tgt
        n_pad = (nq + 63) // 64 * 64
        vt = ops._workspace(f"upn_dec_vt_{C}x{n_pad}", self.dev, C * n_pad * 2)[:C * n_pad * 2].view(BF).view(C, n_pad)
        hd = C // H
        for ly in self.layers:
            qpos = self.ref_head(ops.sine_embed(ref, 4))                             # conditional query position from the current boxes
            q = ops.add(out, qpos)
            qk = ops.gemm(q, ly["w_qk"], ly["b_qk"])                                # self-attention: q = k = tgt + pos, v = tgt
            v = ops.gemm(out, ly["w_v"], ly["b_v"])
            ops.transpose_into(v, vt, 0)
            att = ops.attention(qk[:, :C], qk[:, C:], vt, self.items, H, H, hd, float(hd) ** -0.5, False, flops=4.0 * C * nq * nq)
            out = ops.layernorm(ops.gemm(att, ly["w_o"], ly["b_o"], residual=out), ly["n2"][0], ly["n2"][1], 1e-5)
            q = ops.add(out, qpos)                                                   # cross-attention into the image memory
            y = ly["cross"].attend(q, ref.view(1, nq, 1, 4), memory, shapes_dev, start_dev, residual=out)
            out = ops.layernorm(y, ly["n1"][0], ly["n1"][1], 1e-5)
            h = ops.gemm(out, ly["w1"], ly["b1"], act=ops.ACT_RELU)
            out = ops.layernorm(ops.gemm(h, ly["w2"], ly["b2"], residual=out), ly["n3"][0], ly["n3"][1], 1e-5)
            ref = ops.box_refine(self.bbox(out, last_f32=True), ref, mode=0)         # iterative box refinement
            refs.append(ref)
            hs.append(ops.layernorm(out, self.norm[0], self.norm[1], 1e-5))
        boxes = ops.box_refine(self.bbox(hs[-1], last_f32=True), refs[-2], mode=0)
        logits = ops.gemm(hs[-1], self.prompts[prompt_type], out_f32=True)
        return dict(hs=hs, refs=refs, pred_boxes=boxes, pred_logits=logits[:, 0])


# ---- Swin-L backbone ----------------------------------------------------------------------------------------------------------------
# Reference: models/backbone/swin.py (PatchEmbed :484-522, SwinTransformerBlock :259-318, WindowAttention :136-175, PatchMerging
# :333-357, BasicLayer :440-481, SwinTransformer.forward :700-744).  Same launch shape per block as the DaViT spatial block of the
# main path: LayerNorm -> pad + cyclic shift + window partition (one gather) -> qkv GEMM -> window attention with the relative-position
# bias and the shift mask inside the MFMA kernel -> proj GEMM -> reverse + un-shift + crop + residual (one gather) -> LayerNorm ->
# fc1 + GELU -> fc2 + residual.  The bias table is expanded to a dense fp32 [heads, 144, 144] per block once at load.
def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _rel_pos_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


class SwinBackbone:
    def __init__(self, state: Dict[str, torch.Tensor], depths: Sequence[int], heads: Sequence[int], window: int, device="cuda",
                 prefix: str = "backbone.model.backbone."):
        self.dev = torch.device(device)
        self.depths, self.heads, self.ws = list(depths), list(heads), window
        w = state[prefix + "patch_embed.proj.weight"]                       # [C0, 3, 4, 4]
        self.C0 = w.shape[0]
        wp = torch.zeros(self.C0, 4, 4, 8, dtype=w.dtype)                   # im2col order [ky][kx][c], image staged with 8 channels
        wp[..., :3] = w.permute(0, 2, 3, 1)
        self.pe_w, self.pe_b = _dev(wp.reshape(self.C0, 128), self.dev), _dev(state[prefix + "patch_embed.proj.bias"], self.dev)
        self.pe_n = (_dev(state[prefix + "patch_embed.norm.weight"], self.dev), _dev(state[prefix + "patch_embed.norm.bias"], self.dev))
        rel = _rel_pos_index(window).view(-1)
        wl = window * window
        self.stages = []
        for i, depth in enumerate(self.depths):
            blocks = []
            for j in range(depth):
                p = f"{prefix}layers.{i}.blocks.{j}."
                bias = state[p + "attn.relative_position_bias_table"].float()[rel].view(wl, wl, -1).permute(2, 0, 1).contiguous()
                blocks.append(dict(
                    n1=(_dev(state[p + "norm1.weight"], self.dev), _dev(state[p + "norm1.bias"], self.dev)),
                    n2=(_dev(state[p + "norm2.weight"], self.dev), _dev(state[p + "norm2.bias"], self.dev)),
                    qkv_w=_dev(state[p + "attn.qkv.weight"], self.dev), qkv_b=_dev(state[p + "attn.qkv.bias"], self.dev),
                    proj_w=_dev(state[p + "attn.proj.weight"], self.dev), proj_b=_dev(state[p + "attn.proj.bias"], self.dev),
                    fc1_w=_dev(state[p + "mlp.fc1.weight"], self.dev), fc1_b=_dev(state[p + "mlp.fc1.bias"], self.dev),
                    fc2_w=_dev(state[p + "mlp.fc2.weight"], self.dev), fc2_b=_dev(state[p + "mlp.fc2.bias"], self.dev),
                    bias=bias.to(self.dev), shift=0 if j % 2 == 0 else window // 2))
            st = dict(blocks=blocks, norm=(_dev(state[f"{prefix}norm{i}.weight"], self.dev), _dev(state[f"{prefix}norm{i}.bias"], self.dev)))
            if i < len(self.depths) - 1:
                p = f"{prefix}layers.{i}.downsample."
                st["merge"] = dict(n=(_dev(state[p + "norm.weight"], self.dev), _dev(state[p + "norm.bias"], self.dev)), w=_dev(state[p + "reduction.weight"], self.dev))
            self.stages.append(st)
        self._items: Dict[tuple, torch.Tensor] = {}

    def _window_items(self, n_windows: int, heads: int):
        key = (n_windows, heads)
        if key not in self._items:
            wl = self.ws * self.ws
            segs = [(i * wl, (i + 1) * wl) for i in range(n_windows)]
            self._items[key] = ops.make_items(segs, self.dev, block=ops.pick_q_block(segs, heads))
        return self._items[key]

    def _block(self, x, H, W, C, heads, b):
        ws, shift = self.ws, b["shift"]
        h = ops.layernorm(x, b["n1"][0], b["n1"][1], 1e-5)
        hw = ops.swin_window_partition(h, H, W, ws, shift)
        qkv = ops.gemm(hw, b["qkv_w"], b["qkv_b"])
        n = hw.shape[0]
        n_pad = _round_up(n, 64)
        vt = ops._workspace(f"swin_vt_{C}x{n_pad}", self.dev, C * n_pad * 2)[:C * n_pad * 2].view(BF).view(C, n_pad)
        ops.transpose_into(qkv[:, 2 * C:], vt, 0)
        hd = C // heads
        nwy, nwx = -(-H // ws), -(-W // ws)
        att = ops.attention_window_bias(qkv[:, :C], qkv[:, C:2 * C], vt, self._window_items(nwy * nwx, heads), heads, hd, float(hd) ** -0.5, b["bias"], ws, shift,
                                        nwy, nwx, flops=4.0 * C * n * ws * ws)
        y = ops.gemm(att, b["proj_w"], b["proj_b"])
        x = ops.swin_window_reverse_add(y, x, H, W, ws, shift)
        h = ops.layernorm(x, b["n2"][0], b["n2"][1], 1e-5)
        h = ops.gemm(h, b["fc1_w"], b["fc1_b"], act=ops.ACT_GELU)
        return ops.gemm(h, b["fc2_w"], b["fc2_b"], residual=x)

    def forward(self, img: torch.Tensor):
        """img [3, H, W] on the device (normalised, fp32 or bf16) -> ([token-major normed stage outputs bf16 [H_l*W_l, C_l]], [(H_l, W_l)])."""
        if not img.is_cuda or img.dim() != 3:
            raise TypeError("SwinBackbone: image must be a [3, H, W] GPU tensor (no CPU path exists)")
        _, H, W = img.shape
        Hp, Wp = _round_up(H, 4), _round_up(W, 4)
        if (Hp, Wp) != (H, W):                                  # PatchEmbed pads right / bottom with zeros to the patch multiple (:506-509)
            pad = torch.zeros(3, Hp, Wp, dtype=img.dtype, device=img.device)
            pad[:, :H, :W].copy_(img)
            img = pad
        x = ops.nchw_to_hwc8(img.contiguous())
        col, H, W = ops.im2col(x, Hp, Wp, 4, 4, 4, 0)
        x = ops.layernorm(ops.gemm(col, self.pe_w, self.pe_b), self.pe_n[0], self.pe_n[1], 1e-5)
        feats, sizes = [], []
        C = self.C0
        for i, st in enumerate(self.stages):
            for b in st["blocks"]:
                x = self._block(x, H, W, C, self.heads[i], b)
            feats.append(ops.layernorm(x, st["norm"][0], st["norm"][1], 1e-5))
            sizes.append((H, W))
            if "merge" in st:
                g = ops.patch_merge(x, H, W)
                H, W = (H + 1) // 2, (W + 1) // 2
                x = ops.gemm(ops.layernorm(g, st["merge"]["n"][0], st["merge"]["n"][1], 1e-5), st["merge"]["w"])
                C *= 2
        return feats, sizes


def position_embedding_sine_hw(H: int, W: int, num_pos_feats: int = 128, temp_h: float = 20.0, temp_w: float = 20.0) -> torch.Tensor:
    """PositionEmbeddingSineHW (utils/detr_utils.py:110-148; normalize=True, temperatures 20 in configs/upn_large.py) on an unpadded map:
    fp32 [H*W, 2*num_pos_feats] (pos_y | pos_x), computed on the host (a function of the shape only)."""
    scale, eps = 2 * np.pi, 1e-6
    y = np.arange(1, H + 1, dtype=np.float32)[:, None].repeat(W, 1)
    x = np.arange(1, W + 1, dtype=np.float32)[None, :].repeat(H, 0)
    y = y / (y[-1:, :] + np.float32(eps)) * np.float32(scale)
    x = x / (x[:, -1:] + np.float32(eps)) * np.float32(scale)
    d = np.arange(num_pos_feats, dtype=np.float32)
    dx = np.float32(temp_w) ** (2 * (d // 2) / np.float32(num_pos_feats))
    dy = np.float32(temp_h) ** (2 * (d // 2) / np.float32(num_pos_feats))
    px, py = x[:, :, None] / dx, y[:, :, None] / dy
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), 3).reshape(H, W, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), 3).reshape(H, W, -1)
    return torch.from_numpy(np.concatenate((py, px), 2).reshape(H * W, -1).astype(np.float32))


class InputProjection:
    """upn_model.py:143-216: 1x1 conv + GroupNorm(32) per backbone level, 3x3 stride-2 conv + GroupNorm for the extra level, then the
    level-major flatten with (sine position + level) embeddings."""

    def __init__(self, state, device="cuda", n_levels: int = 5, groups: int = 32):
        self.dev = torch.device(device)
        self.groups, self.n_levels = groups, n_levels
        self.proj = []
        l = 0
        while f"input_proj.{l}.0.weight" in state:
            w = state[f"input_proj.{l}.0.weight"]
            k = w.shape[-1]
            wg = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)                # [Cout][ky][kx][Cin] (im2col order)
            self.proj.append(dict(k=k, w=_dev(wg, self.dev), b=_dev(state[f"input_proj.{l}.0.bias"], self.dev),
                                  gw=_dev(state[f"input_proj.{l}.1.weight"], self.dev), gb=_dev(state[f"input_proj.{l}.1.bias"], self.dev)))
            l += 1
        self.level_embed = state["transformer.level_embed"].float()
        self._pos: Dict[tuple, torch.Tensor] = {}

    def forward(self, feats: List[torch.Tensor], sizes: List[Tuple[int, int]]):
        srcs, shapes = [], []
        for l, (f, (H, W)) in enumerate(zip(feats, sizes)):
            pr = self.proj[l]
            srcs.append(ops.groupnorm_tokens(ops.gemm(f, pr["w"], pr["b"]), self.groups, pr["gw"], pr["gb"], 1e-5))
            shapes.append((H, W))
        for l in range(len(feats), self.n_levels):
            pr = self.proj[l]
            inp, (H, W) = (feats[-1], sizes[-1]) if l == len(feats) else (srcs[-1], shapes[-1])
            col, Ho, Wo = ops.im2col(inp, H, W, 3, 3, 2, 1)
            srcs.append(ops.groupnorm_tokens(ops.gemm(col, pr["w"], pr["b"]), self.groups, pr["gw"], pr["gb"], 1e-5))
            shapes.append((Ho, Wo))
        key = tuple(shapes)
        if key not in self._pos:
            pos = torch.cat([position_embedding_sine_hw(H, W) + self.level_embed[l][None] for l, (H, W) in enumerate(shapes)], 0)
            self._pos[key] = pos.to(device=self.dev, dtype=BF).contiguous()
        S = sum(h * w for h, w in shapes)
        src = torch.empty(S, srcs[0].shape[1], dtype=BF, device=self.dev)    # level-major flatten (row copies into one buffer)
        r = 0
        for s in srcs:
            src[r:r + s.shape[0]].copy_(s)
            r += s.shape[0]
        return src, self._pos[key], shapes


class UPNEngine:
    """The whole UPN forward for one image: Swin-L -> input projections -> deformable encoder -> query selection -> decoder -> heads
    (models/architecture/upn_model.py:86-140).  configs/upn_large.py: depths [2, 2, 18, 2], 6 + 6 layers, 900 queries."""

    def __init__(self, state, device="cuda", depths=(2, 2, 18, 2), heads=(6, 12, 24, 48), window: int = 12, n_enc: int = 6, n_dec: int = 6,
                 n_queries: int = 900):
        self.backbone = SwinBackbone(state, depths, heads, window, device)
        self.proj = InputProjection(state, device)
        self.encoder = DeformableEncoder(state, "transformer.encoder.", n_enc, device)
        self.selector = QuerySelector(state, device, n_queries)
        self.decoder = DeformableDecoder(state, n_dec, device, n_queries)
        import collections
        self._graphs = collections.OrderedDict()     # (image shape, prompt type) -> (hipGraph, static image, outputs): LRU of GRAPH_CACHE
        self._seen = {}
        self._ws_owner = ops.new_owner(self)         # scratch buffers are keyed by this token (ops.workspace_scope)

    GRAPH_CACHE = 4
    CAPTURE_AFTER = 1       # a size is captured on its second sighting (the first call also builds the per-size host plans)

    def forward_graph(self, img: torch.Tensor, prompt_type: str = "fine_grained_prompt") -> dict:
        """forward() replayed as ONE hipGraph per (image size, prompt type): the detector is ~1 900 small launches per image, most of
        them launch-bound when issued one by one.  Returns pred_boxes / pred_logits (static buffers of the graph: valid until the
        next call for the same size).  Same kernels, same order: bit-identical to forward() (tests/test_upn_gpu.py)."""
        key = (tuple(img.shape), prompt_type)
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen.get(key, 0)
            if len(self._seen) >= 4096:                  # bounded: a dataset has thousands of distinct image sizes
                self._seen.pop(next(iter(self._seen)))
            self._seen[key] = n + 1
            if n < self.CAPTURE_AFTER:
                with ops.workspace_scope(self._ws_owner):
                    return self.forward(img, prompt_type)
            with ops.graph_lock.capture(), torch.inference_mode(False), ops.workspace_scope(self._ws_owner):
                static = img.clone()
                side = torch.cuda.Stream()           # warm-up on a side stream (allocates every lazily-created scratch buffer), then capture
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.forward(static, prompt_type)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = self.forward(static, prompt_type)
                ent = (g, static, dict(pred_boxes=out["pred_boxes"], pred_logits=out["pred_logits"]))
                self._graphs[key] = ent
                while len(self._graphs) > self.GRAPH_CACHE:
                    self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        g, static, out = ent
        static.copy_(img)
        with ops.graph_lock.replay():
            g.replay()
        return out

    def forward(self, img: torch.Tensor, prompt_type: str = "fine_grained_prompt") -> dict:
        if prompt_type not in ("fine_grained_prompt", "coarse_grained_prompt"):
            raise ValueError("prompt_type must be 'fine_grained_prompt' or 'coarse_grained_prompt' (inference_wrapper.py:51-52)")
        feats, sizes = self.backbone.forward(img)
        src, pos, shapes = self.proj.forward(feats, sizes)
        memory = self.encoder.forward(src, pos, shapes)
        sel = self.selector.forward(memory, shapes, prompt_type)
        out = self.decoder.forward(memory, shapes, sel["refpoints"], prompt_type)
        return dict(pred_boxes=out["pred_boxes"], pred_logits=out["pred_logits"], feats=feats, sizes=sizes, src=src, pos=pos, shapes=shapes,
                    memory=memory, selection=sel)
