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


# ---- query selection, decoder, heads ---------------------------------------------------------------------------------------------
# models/utils/detr_utils.py:269-273 (inverse_sigmoid), :276-310 (gen_sineembed_for_position), :351-415 (gen_encoder_output_proposals)
# models/architecture/deformable_transformer.py:262-336 (get_two_stage_proposal), models/decoder/upn_decoder.py:98-139, :262-378
# models/architecture/upn_model.py:96-140 (prediction heads), models/module/contrastive.py (ContrastiveAssign), models/module/mlp.py
import math


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def sine_embed(pos_tensor):
    scale = 2 * math.pi
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / 128)

    def one(v):
        p = v[..., None] * scale / dim_t
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)

    parts = [one(pos_tensor[..., 1]), one(pos_tensor[..., 0])]
    if pos_tensor.shape[-1] == 4:
        parts += [one(pos_tensor[..., 2]), one(pos_tensor[..., 3])]
    return torch.cat(parts, -1)


def mlp(state, prefix, x, n):
    for i in range(n):
        x = F.linear(x, state[f"{prefix}layers.{i}.weight"], state[f"{prefix}layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def encoder_output_proposals(shapes):
    """gen_encoder_output_proposals for one unpadded image: (keep mask [S] bool, proposals in logit space [S, 4], +inf where invalid)."""
    props = []
    for lvl, (H, W) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([W, H], dtype=torch.float32)
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(-1, 4))
    p = torch.cat(props, 0)
    valid = ((p > 0.01) & (p < 0.99)).all(-1)
    logit = torch.log(p / (1 - p))
    logit = logit.masked_fill(~valid[:, None], float("inf"))
    return valid, logit


def query_selection(state, memory, shapes, n_queries, prompt="fine_grained_prompt"):
    """-> (scores [S], coords_unsig [S, 4], topk indices [nq], refpoints_unsig [nq, 4]); memory [S, C]."""
    valid, props = encoder_output_proposals(shapes)
    om = memory * valid[:, None].to(memory.dtype)
    om = F.layer_norm(F.linear(om, state["transformer.enc_output.weight"], state["transformer.enc_output.bias"]), (om.shape[-1],),
                      state["transformer.enc_output_norm.weight"], state["transformer.enc_output_norm.bias"], 1e-5)
    scores = om @ state[f"transformer.{prompt}.weight"][0]
    coords = mlp(state, "transformer.enc_out_bbox_embed.", om, 3) + props
    idx = torch.topk(scores, n_queries)[1]
    return scores, coords, idx, coords[idx]


def decoder(state, memory, shapes, refpoints_unsig, n_layers, n_heads=8, prompt="fine_grained_prompt"):
    """UPNDecoder.forward + the UPN heads for one image -> (hs [n_layers, nq, C] normed, refs [n_layers + 1, nq, 4], pred_boxes, pred_logits)."""
    C = memory.shape[-1]
    level_start = [0]
    for h, w in shapes[:-1]:
        level_start.append(level_start[-1] + h * w)
    L = len(shapes)
    tgt = state["transformer.tgt_embed.weight"]
    ref = refpoints_unsig.sigmoid()
    hs, refs = [], [ref]
    out = tgt
    for i in range(n_layers):
        p = f"transformer.decoder.layers.{i}."
        ref_in = ref[:, None, :].expand(-1, L, -1)                                   # valid ratios are 1
        qpos = mlp(state, "transformer.decoder.ref_point_head.", sine_embed(ref_in[:, 0, :]), 2)
        q = k = out + qpos
        w, b = state[p + "self_attn.in_proj_weight"], state[p + "self_attn.in_proj_bias"]
        qh = F.linear(q, w[:C], b[:C]).view(-1, n_heads, C // n_heads).transpose(0, 1)
        kh = F.linear(k, w[C:2 * C], b[C:2 * C]).view(-1, n_heads, C // n_heads).transpose(0, 1)
        vh = F.linear(out, w[2 * C:], b[2 * C:]).view(-1, n_heads, C // n_heads).transpose(0, 1)
        att = torch.softmax(qh @ kh.transpose(1, 2) * (C // n_heads) ** -0.5, -1) @ vh
        att = F.linear(att.transpose(0, 1).reshape(-1, C), state[p + "self_attn.out_proj.weight"], state[p + "self_attn.out_proj.bias"])
        out = F.layer_norm(out + att, (C,), state[p + "norm2.weight"], state[p + "norm2.bias"], 1e-5)
        ca = ms_deform_attn(state, p + "cross_attn.", (out + qpos)[None], ref_in[None].contiguous(), memory[None], shapes, level_start)[0]
        out = F.layer_norm(out + ca, (C,), state[p + "norm1.weight"], state[p + "norm1.bias"], 1e-5)
        h = F.relu(F.linear(out, state[p + "linear1.weight"], state[p + "linear1.bias"]))
        out = F.layer_norm(out + F.linear(h, state[p + "linear2.weight"], state[p + "linear2.bias"]), (C,), state[p + "norm3.weight"], state[p + "norm3.bias"], 1e-5)
        ref = (mlp(state, "bbox_embed.0.", out, 3) + inverse_sigmoid(ref)).sigmoid()
        refs.append(ref)
        hs.append(F.layer_norm(out, (C,), state["transformer.decoder.norm.weight"], state["transformer.decoder.norm.bias"], 1e-5))
    boxes = (mlp(state, "bbox_embed.0.", hs[-1], 3) + inverse_sigmoid(refs[-2])).sigmoid()
    logits = hs[-1] @ state[f"transformer.{prompt}.weight"].t()
    return torch.stack(hs), torch.stack(refs), boxes, logits


# ---- Swin backbone, position embedding, input projections ---------------------------------------------------------------------------
# models/backbone/swin.py: PatchEmbed :484-522, SwinTransformerBlock.forward :259-318, WindowAttention.forward :136-175,
# PatchMerging.forward :333-357, BasicLayer.forward :440-481 (shift mask), SwinTransformer.forward :700-744
# models/utils/detr_utils.py:110-148 (PositionEmbeddingSineHW), models/architecture/upn_model.py:143-216 (forward_backbone_encoder)
def _rel_pos_index(ws):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(Hp, Wp, ws, shift):
    img = torch.zeros(Hp, Wp)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    m = mw[:, None, :] - mw[:, :, None]
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_block(state, p, x, H, W, heads, ws, shift, rel_index):
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), state[p + "norm1.weight"], state[p + "norm1.bias"], 1e-5).view(H, W, C)
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    h = F.pad(h, (0, 0, 0, Wp - W, 0, Hp - H))
    if shift:
        h = torch.roll(h, (-shift, -shift), (0, 1))
    win = h.view(Hp // ws, ws, Wp // ws, ws, C).permute(0, 2, 1, 3, 4).reshape(-1, ws * ws, C)
    qkv = F.linear(win, state[p + "attn.qkv.weight"], state[p + "attn.qkv.bias"]).view(-1, ws * ws, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    att = (qkv[0] * (C // heads) ** -0.5) @ qkv[1].transpose(-2, -1)
    att = att + state[p + "attn.relative_position_bias_table"][rel_index.view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1)[None]
    if shift:
        att = att + _shift_mask(Hp, Wp, ws, shift)[:, None]
    o = (torch.softmax(att, -1) @ qkv[2]).transpose(1, 2).reshape(-1, ws * ws, C)
    o = F.linear(o, state[p + "attn.proj.weight"], state[p + "attn.proj.bias"])
    o = o.view(Hp // ws, Wp // ws, ws, ws, C).permute(0, 2, 1, 3, 4).reshape(Hp, Wp, C)
    if shift:
        o = torch.roll(o, (shift, shift), (0, 1))
    x = x + o[:H, :W].reshape(H * W, C)
    h = F.layer_norm(x, (C,), state[p + "norm2.weight"], state[p + "norm2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, state[p + "mlp.fc1.weight"], state[p + "mlp.fc1.bias"])), state[p + "mlp.fc2.weight"], state[p + "mlp.fc2.bias"])
    return x + h


def swin_forward(state, img, depths, heads, ws, prefix="backbone.model.backbone."):
    """img [3, H, W] -> ([token-major normed stage outputs [H_l*W_l, C_l]], [(H_l, W_l)])."""
    _, H, W = img.shape
    x = F.pad(img, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))[None]
    x = F.conv2d(x, state[prefix + "patch_embed.proj.weight"], state[prefix + "patch_embed.proj.bias"], stride=4)
    H, W = x.shape[2:]
    x = x[0].flatten(1).t()
    x = F.layer_norm(x, (x.shape[-1],), state[prefix + "patch_embed.norm.weight"], state[prefix + "patch_embed.norm.bias"], 1e-5)
    rel = _rel_pos_index(ws)
    outs, sizes = [], []
    for i, depth in enumerate(depths):
        C = x.shape[-1]
        for j in range(depth):
            x = swin_block(state, f"{prefix}layers.{i}.blocks.{j}.", x, H, W, heads[i], ws, 0 if j % 2 == 0 else ws // 2, rel)
        outs.append(F.layer_norm(x, (C,), state[f"{prefix}norm{i}.weight"], state[f"{prefix}norm{i}.bias"], 1e-5))
        sizes.append((H, W))
        if i < len(depths) - 1:
            g = F.pad(x.view(H, W, C), (0, 0, 0, W % 2, 0, H % 2))
            g = torch.cat([g[0::2, 0::2], g[1::2, 0::2], g[0::2, 1::2], g[1::2, 1::2]], -1)
            H, W = g.shape[:2]
            g = g.reshape(H * W, 4 * C)
            p = f"{prefix}layers.{i}.downsample."
            x = F.linear(F.layer_norm(g, (4 * C,), state[p + "norm.weight"], state[p + "norm.bias"], 1e-5), state[p + "reduction.weight"])
    return outs, sizes


def position_embedding(H, W, num_pos_feats=128, temp_h=20, temp_w=20):
    """PositionEmbeddingSineHW with normalize=True on an unpadded H x W map -> [H*W, 2*num_pos_feats] (pos_y | pos_x)."""
    scale, eps = 2 * math.pi, 1e-6
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    dx = temp_w ** (2 * (d // 2) / num_pos_feats)
    dy = temp_h ** (2 * (d // 2) / num_pos_feats)
    px, py = x[:, :, None] / dx, y[:, :, None] / dy
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), 3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), 3).flatten(2)
    return torch.cat((py, px), 2).reshape(H * W, -1)


def backbone_encoder_inputs(state, feats, sizes, n_levels=5, groups=32):
    """input_proj (1x1 conv + GroupNorm per backbone level, 3x3 stride-2 conv + GroupNorm for the extra level), sine position
    embedding + level embedding, flattened level-major -> (src [S, 256], pos [S, 256], shapes)."""
    srcs, shapes = [], []
    for l, (f, (H, W)) in enumerate(zip(feats, sizes)):
        m = f.t().reshape(1, -1, H, W)
        s = F.group_norm(F.conv2d(m, state[f"input_proj.{l}.0.weight"], state[f"input_proj.{l}.0.bias"]), groups, state[f"input_proj.{l}.1.weight"],
                         state[f"input_proj.{l}.1.bias"], 1e-5)
        srcs.append(s)
        shapes.append((H, W))
    for l in range(len(feats), n_levels):
        inp = feats[-1].t().reshape(1, -1, *sizes[-1]) if l == len(feats) else srcs[-1]
        s = F.group_norm(F.conv2d(inp, state[f"input_proj.{l}.0.weight"], state[f"input_proj.{l}.0.bias"], stride=2, padding=1), groups,
                         state[f"input_proj.{l}.1.weight"], state[f"input_proj.{l}.1.bias"], 1e-5)
        srcs.append(s)
        shapes.append(tuple(s.shape[2:]))
    src = torch.cat([s[0].flatten(1).t() for s in srcs], 0)
    pos = torch.cat([position_embedding(H, W) + state["transformer.level_embed"][l][None] for l, (H, W) in enumerate(shapes)], 0)
    return src, pos, shapes


def detect(state, img, depths, heads, ws, n_enc, n_dec, n_queries, prompt="fine_grained_prompt"):
    """The whole UPN forward for one image -> (pred_boxes [nq, 4] cxcywh in [0, 1], pred_logits [nq])."""
    feats, sizes = swin_forward(state, img, depths, heads, ws)
    src, pos, shapes = backbone_encoder_inputs(state, feats, sizes)
    enc_state = {k[len("transformer.encoder."):]: v for k, v in state.items() if k.startswith("transformer.encoder.")}
    memory = encoder(enc_state, src[None], pos[None], shapes, n_enc)[0]
    _, _, _, refp = query_selection(state, memory, shapes, n_queries, prompt)
    _, _, boxes, logits = decoder(state, memory, shapes, refp, n_dec, prompt=prompt)
    return boxes, logits[:, 0]
