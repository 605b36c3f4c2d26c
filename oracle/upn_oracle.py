"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement (torch fp32) of the UPN proposal detector's deformable-transformer stages, pinned against the reference's own
modules through tests/golden/upn_ref.npz (tests/golden/make_upn_golden.py runs detect_tools/upn/models/* in place).  Follows
  ops/modules/ms_deform_attn.py:100-204          MSDeformAttn.forward (value_proj, sampling_offsets, softmax(attention_weights),
                                                 sampling locations for 2-d / 4-d reference points, MSDA, output_proj)
  models/encoder/upn_encoder.py:62-110,198-213   DeformableTransformerEncoderLayer.forward, UPNEncoder.get_reference_points
The MSDA operator itself is oracle/msda_ref.c (pinned separately)."""
import torch
import torch.nn.functional as F

from . import msda_oracle


def ms_deform_attn(state, prefix, query, reference_points, input_flatten, shapes, level_start, n_heads=8, n_points=4):
    """query [N, Lq, C], reference_points [N, Lq, L, 2|4], input_flatten [N, S, C] -> [N, Lq, C]."""
    N, Lq, C = query.shape
    S = input_flatten.shape[1]
    L, M, P = len(shapes), n_heads, n_points
    value = F.linear(input_flatten, state[prefix + "value_proj.weight"], state[prefix + "value_proj.bias"]).view(N, S, M, C // M)
    off = F.linear(query, state[prefix + "sampling_offsets.weight"], state[prefix + "sampling_offsets.bias"]).view(N, Lq, M, L, P, 2)
    aw = F.linear(query, state[prefix + "attention_weights.weight"], state[prefix + "attention_weights.bias"]).view(N, Lq, M, L * P)
    aw = F.softmax(aw, -1).view(N, Lq, M, L, P)
    sh = torch.as_tensor(shapes, dtype=torch.float32)
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([sh[:, 1], sh[:, 0]], -1)                              # (W_l, H_l)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
    out = msda_oracle.ms_deform_attn_forward(value.contiguous(), shapes, level_start, loc.contiguous(), aw.contiguous())
    return F.linear(out, state[prefix + "output_proj.weight"], state[prefix + "output_proj.bias"])


def encoder_reference_points(shapes, valid_ratios=None):
    """UPNEncoder.get_reference_points (upn_encoder.py:198-213) for one image: [1, S, L, 2]."""
    L = len(shapes)
    vr = torch.ones(1, L, 2) if valid_ratios is None else valid_ratios
    pts = []
    for lvl, (H, W) in enumerate(shapes):
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ref_y = ref_y.reshape(-1)[None] / (vr[:, None, lvl, 1] * H)
        ref_x = ref_x.reshape(-1)[None] / (vr[:, None, lvl, 0] * W)
        pts.append(torch.stack((ref_x, ref_y), -1))
    ref = torch.cat(pts, 1)
    return ref[:, :, None] * vr[:, None]


def encoder_layer(state, prefix, src, pos, ref, shapes, level_start):
    src2 = ms_deform_attn(state, prefix + "self_attn.", src + pos, ref, src, shapes, level_start)
    src = F.layer_norm(src + src2, (src.shape[-1],), state[prefix + "norm1.weight"], state[prefix + "norm1.bias"], 1e-5)
    h = F.relu(F.linear(src, state[prefix + "linear1.weight"], state[prefix + "linear1.bias"]))
    src2 = F.linear(h, state[prefix + "linear2.weight"], state[prefix + "linear2.bias"])
    return F.layer_norm(src + src2, (src.shape[-1],), state[prefix + "norm2.weight"], state[prefix + "norm2.bias"], 1e-5)


def encoder(state, src, pos, shapes, n_layers, collect=None):
    """UPNEncoder.forward (upn_encoder.py:215-288; no fusion layers in configs/upn_large.py), one image, no padding."""
    level_start = [0]
    for h, w in shapes[:-1]:
        level_start.append(level_start[-1] + h * w)
    ref = encoder_reference_points(shapes)
    out = src
    for i in range(n_layers):
        out = encoder_layer(state, f"layers.{i}.", out, pos, ref, shapes, level_start)
        if collect is not None:
            collect.append(out)
    return out
