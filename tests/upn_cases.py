"""Seeded weights and inputs of the UPN (proposal detector) cases — shared by tests/golden/make_upn_golden.py, which feeds them to the
reference's own modules imported in place, and by the oracle / GPU tests, which regenerate them bit for bit (CPU generator, same
torch build on both boxes).  All values are bf16-representable fp32, so the engine's bf16 weights ARE the reference's fp32 weights."""
import math

import torch

D_MODEL, N_HEADS, N_LEVELS, N_POINTS, D_FFN = 256, 8, 5, 4, 2048     # reference configs/upn_large.py
ENC_SHAPES = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]              # a small 5-level pyramid (S = 257)


def _bf(t):
    return t.bfloat16().float()


def level_start(shapes):
    out = [0]
    for h, w in shapes[:-1]:
        out.append(out[-1] + h * w)
    return out


def msda_module_state(g, prefix, d=D_MODEL, M=N_HEADS, L=N_LEVELS, P=N_POINTS):
    """State of one MSDeformAttn (reference ops/modules/ms_deform_attn.py:70-73): random projections + the reference's own
    ring-shaped offset bias (:77-90) so that the sampling points spread over several pixels."""
    thetas = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    for i in range(P):
        grid[:, :, i, :] *= i + 1
    s = {
        prefix + "sampling_offsets.weight": torch.randn(M * L * P * 2, d, generator=g) * 0.02,
        prefix + "sampling_offsets.bias": grid.reshape(-1) + torch.randn(M * L * P * 2, generator=g) * 0.1,
        prefix + "attention_weights.weight": torch.randn(M * L * P, d, generator=g) * 0.05,
        prefix + "attention_weights.bias": torch.randn(M * L * P, generator=g) * 0.2,
        prefix + "value_proj.weight": torch.randn(d, d, generator=g) / 16.0,
        prefix + "value_proj.bias": torch.randn(d, generator=g) * 0.05,
        prefix + "output_proj.weight": torch.randn(d, d, generator=g) / 16.0,
        prefix + "output_proj.bias": torch.randn(d, generator=g) * 0.05,
    }
    return {k: _bf(v) for k, v in s.items()}


def _norm_state(g, prefix, d):
    return {prefix + "weight": _bf(1.0 + 0.1 * torch.randn(d, generator=g)), prefix + "bias": _bf(0.1 * torch.randn(d, generator=g))}


def encoder_state(n_layers=2, seed=4242, d=D_MODEL, d_ffn=D_FFN):
    """State dict of the reference's UPNEncoder (models/encoder/upn_encoder.py): layers.{i}.{self_attn.*, norm1, linear1, linear2, norm2}."""
    g = torch.Generator().manual_seed(seed)
    s = {}
    for i in range(n_layers):
        p = f"layers.{i}."
        s.update(msda_module_state(g, p + "self_attn."))
        s.update(_norm_state(g, p + "norm1.", d))
        s[p + "linear1.weight"] = _bf(torch.randn(d_ffn, d, generator=g) / 16.0)
        s[p + "linear1.bias"] = _bf(torch.randn(d_ffn, generator=g) * 0.05)
        s[p + "linear2.weight"] = _bf(torch.randn(d, d_ffn, generator=g) / 45.0)
        s[p + "linear2.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
        s.update(_norm_state(g, p + "norm2.", d))
    return s


def encoder_inputs(shapes=ENC_SHAPES, seed=77, d=D_MODEL):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    src = _bf(torch.randn(1, S, d, generator=g))
    pos = _bf(torch.randn(1, S, d, generator=g) * 0.5)
    return src, pos


# ---- the whole deformable transformer + heads (models/architecture/deformable_transformer.py, upn_model.py) ----------------------
N_QUERIES_SMALL = 30


def _mlp_state(g, prefix, dims, scale=None):
    s = {}
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        s[f"{prefix}layers.{i}.weight"] = _bf(torch.randn(b, a, generator=g) * (scale or 1.0 / math.sqrt(a)))
        s[f"{prefix}layers.{i}.bias"] = _bf(torch.randn(b, generator=g) * 0.05)
    return s


def transformer_state(n_enc=2, n_dec=2, n_queries=N_QUERIES_SMALL, seed=777, d=D_MODEL, d_ffn=D_FFN):
    """State of the reference's UPN model minus backbone / input_proj (keys relative to the UPN module)."""
    g = torch.Generator().manual_seed(seed)
    s = {"transformer.encoder." + k: v for k, v in encoder_state(n_enc, seed + 1).items()}
    for i in range(n_dec):
        p = f"transformer.decoder.layers.{i}."
        s.update(msda_module_state(g, p + "cross_attn."))
        s.update(_norm_state(g, p + "norm1.", d))
        s[p + "self_attn.in_proj_weight"] = _bf(torch.randn(3 * d, d, generator=g) / 16.0)
        s[p + "self_attn.in_proj_bias"] = _bf(torch.randn(3 * d, generator=g) * 0.05)
        s[p + "self_attn.out_proj.weight"] = _bf(torch.randn(d, d, generator=g) / 16.0)
        s[p + "self_attn.out_proj.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
        s.update(_norm_state(g, p + "norm2.", d))
        s[p + "linear1.weight"] = _bf(torch.randn(d_ffn, d, generator=g) / 16.0)
        s[p + "linear1.bias"] = _bf(torch.randn(d_ffn, generator=g) * 0.05)
        s[p + "linear2.weight"] = _bf(torch.randn(d, d_ffn, generator=g) / 45.0)
        s[p + "linear2.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
        s.update(_norm_state(g, p + "norm3.", d))
    s.update(_norm_state(g, "transformer.decoder.norm.", d))
    s.update(_mlp_state(g, "transformer.decoder.ref_point_head.", [2 * d, d, d]))
    s.update(_mlp_state(g, "bbox_embed.0.", [d, d, d, 4], scale=0.04))                       # shared by every decoder layer
    s.update(_mlp_state(g, "transformer.enc_out_bbox_embed.", [d, d, d, 4], scale=0.04))     # its own copy (two_stage_bbox_embed_share=False)
    s["transformer.enc_output.weight"] = _bf(torch.randn(d, d, generator=g) / 16.0)
    s["transformer.enc_output.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
    s.update(_norm_state(g, "transformer.enc_output_norm.", d))
    s["transformer.tgt_embed.weight"] = _bf(torch.randn(n_queries, d, generator=g))
    s["transformer.fine_grained_prompt.weight"] = _bf(torch.randn(1, d, generator=g))
    s["transformer.coarse_grained_prompt.weight"] = _bf(torch.randn(1, d, generator=g))
    return s


# ---- Swin-L backbone (true widths, reduced depth) + input projections: the whole detector -----------------------------------------
SWIN_EMBED, SWIN_HEADS, SWIN_WINDOW = 192, [6, 12, 24, 48], 12          # reference swin_L_384_22k (backbone/wrapper.py:286-292)
SWIN_DEPTHS_SMALL = [2, 2, 2, 2]                                         # reference: [2, 2, 18, 2]
IMG_SMALL = (100, 136)                                                   # H, W of the test image: ragged against the patch (4) and the window (12)


def swin_state(depths=SWIN_DEPTHS_SMALL, seed=99, embed=SWIN_EMBED, heads=SWIN_HEADS, ws=SWIN_WINDOW, prefix="backbone.model.backbone."):
    g = torch.Generator().manual_seed(seed)
    s = {prefix + "patch_embed.proj.weight": _bf(torch.randn(embed, 3, 4, 4, generator=g) * 0.15),
         prefix + "patch_embed.proj.bias": _bf(torch.randn(embed, generator=g) * 0.05)}
    s.update(_norm_state(g, prefix + "patch_embed.norm.", embed))
    for i, depth in enumerate(depths):
        C = embed * 2 ** i
        for j in range(depth):
            p = f"{prefix}layers.{i}.blocks.{j}."
            s.update(_norm_state(g, p + "norm1.", C))
            s[p + "attn.relative_position_bias_table"] = _bf(torch.randn((2 * ws - 1) ** 2, heads[i], generator=g) * 0.5)
            s[p + "attn.qkv.weight"] = _bf(torch.randn(3 * C, C, generator=g) / math.sqrt(C))
            s[p + "attn.qkv.bias"] = _bf(torch.randn(3 * C, generator=g) * 0.05)
            s[p + "attn.proj.weight"] = _bf(torch.randn(C, C, generator=g) / math.sqrt(C) * 0.5)
            s[p + "attn.proj.bias"] = _bf(torch.randn(C, generator=g) * 0.05)
            s.update(_norm_state(g, p + "norm2.", C))
            s[p + "mlp.fc1.weight"] = _bf(torch.randn(4 * C, C, generator=g) / math.sqrt(C))
            s[p + "mlp.fc1.bias"] = _bf(torch.randn(4 * C, generator=g) * 0.05)
            s[p + "mlp.fc2.weight"] = _bf(torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C) * 0.5)
            s[p + "mlp.fc2.bias"] = _bf(torch.randn(C, generator=g) * 0.05)
        if i < len(depths) - 1:
            p = f"{prefix}layers.{i}.downsample."
            s[p + "reduction.weight"] = _bf(torch.randn(2 * C, 4 * C, generator=g) / math.sqrt(4 * C))
            s.update(_norm_state(g, p + "norm.", 4 * C))
        s.update(_norm_state(g, f"{prefix}norm{i}.", C))
    return s


def input_proj_state(seed=55, embed=SWIN_EMBED, d=D_MODEL):
    g = torch.Generator().manual_seed(seed)
    s = {}
    chans = [embed * 2 ** i for i in range(4)]
    for l, c in enumerate(chans):
        s[f"input_proj.{l}.0.weight"] = _bf(torch.randn(d, c, 1, 1, generator=g) / math.sqrt(c))
        s[f"input_proj.{l}.0.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
        s.update(_norm_state(g, f"input_proj.{l}.1.", d))
    s["input_proj.4.0.weight"] = _bf(torch.randn(d, chans[-1], 3, 3, generator=g) / math.sqrt(9 * chans[-1]))
    s["input_proj.4.0.bias"] = _bf(torch.randn(d, generator=g) * 0.05)
    s.update(_norm_state(g, "input_proj.4.1.", d))
    s["transformer.level_embed"] = _bf(torch.randn(N_LEVELS, d, generator=g))
    return s


def upn_state(depths=SWIN_DEPTHS_SMALL, n_enc=2, n_dec=2, n_queries=N_QUERIES_SMALL):
    s = transformer_state(n_enc, n_dec, n_queries)
    s.update(swin_state(depths))
    s.update(input_proj_state())
    return s


def test_image(hw=IMG_SMALL, seed=21):
    """Normalised image tensor [3, H, W] (what UPNWrapper.transform_image hands to the model), bf16-representable."""
    g = torch.Generator().manual_seed(seed)
    return _bf(torch.randn(3, hw[0], hw[1], generator=g))
