"""Composition of the CPU oracles (oracle/*.py) into the reference's call chain `encode_images` -> `encode_regions` -> splice ->
`Qwen2_5_VLModel.forward` (omchat_qwen2_5_vl.py:44-128,135-463; modeling_qwen2_5_vl.py:1126-1242) for ANY engine configuration:
with / without SimpleFPN, aux-only, vt-only, `linear` / `mlpNx_gelu` projectors, the no-box dummy, region LayerNorm, the feature-map /
hybrid position embedding, dynamic / squash aux sizes.  Test infrastructure only (imports oracle/)."""
import re

import torch
import torch.nn.functional as F

from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, vit_oracle as VO

FPN_STRIDES = [3.5, 7, 14, 28]
DUMMY_BOX = torch.tensor([[0., 10., 0., 10.]])     # omchat_qwen2_5_vl.py:90-91


def cpu_state(weights):
    return {k: {n: t.float().cpu() for n, t in sd.items()} for k, sd in weights.items()}


def projector(x, sd, prefix, kind):
    """build_vision_projector(_aux) (multimodal_projector/builder.py:39-115) in fp32."""
    if kind == "identity":
        return x
    if kind == "linear":
        return F.linear(x, sd[prefix + "weight"], sd[prefix + "bias"])
    n = int(re.match(r"^mlp(\d+)x_gelu$", kind).group(1))
    for i in range(n):
        x = F.linear(x, sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"])
        if i + 1 < n:
            x = F.gelu(x)
    return x


def vit(sd, pix, gh, gw, vcfg):
    """-> (image tokens before mm_projector [S/4, 2048], captured maps (token-major raster [S,1280]) of every full-attention block)."""
    return VO.vit_forward(sd["vit"], pix.float(), gh, gw, depth=vcfg.depth, n_heads=vcfg.num_heads, fullatt=tuple(vcfg.fullatt_block_indexes))


def davit_maps(sd, aux):
    """aux [3,H,W] -> 4 bf16 NCHW maps (the reference's tower runs in bf16: its outputs are bf16 tensors)."""
    maps, sizes = DO.davit_forward(sd["davit"], aux.float().unsqueeze(0))
    return [m.bfloat16().reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(maps, sizes)]


def nchw(tm, gh, gw):
    return tm.reshape(gh, gw, tm.shape[-1]).permute(2, 0, 1).unsqueeze(0)


def region_features(sd, cfg, aux_nchw, vit_maps, boxes, gh, gw, aux_hw, region_ln=None):
    """encode_regions (:75-128) up to the fp32 HFRE output [N, C_region].  cfg: FO1Config; boxes fp32 [N,4] in aux pixels or None (the
    dummy box); vit_maps: the oracle ViT's captured maps (token-major fp32)."""
    if boxes is None or boxes.shape[0] == 0:
        boxes = DUMMY_BOX
    boxes = boxes.float()
    H, W = aux_hw
    p = cfg.vit.patch_size
    sh, sw = (gh * p) / H, (gw * p) / W
    vtb = boxes * torch.tensor([sw, sh, sw, sh])
    kw = dict(grid_hw=(gh, gw), apply_pos=cfg.mm_apply_position_embedding, region_ln=region_ln, strategy=cfg.mm_pos_embedding_strategy,
              pos_from="aux" if cfg.mm_region_feature_combination == "concat_aux_pos" else "vt")
    if not cfg.mm_use_vision_tower_region_feature:
        return HO.hfre_oracle(aux_nchw, boxes, None, None, region_dim=cfg.mm_region_hidden_size, aux_only=True, **kw)[0]
    vt_only = cfg.mm_use_vt_region_feature_only
    if cfg.mm_use_simpleFPN_for_vt:
        fpn = [m.bfloat16() for m in FO.fpn_forward(sd["fpn"], nchw(vit_maps[-1].bfloat16().float(), gh, gw))]
        return HO.hfre_oracle(aux_nchw, boxes, fpn, vtb, region_dim=cfg.mm_region_hidden_size, vt_strides=FPN_STRIDES, vt_only=vt_only, **kw)[0]
    vt = [nchw(m.bfloat16(), gh, gw) for m in vit_maps]
    return HO.hfre_oracle(aux_nchw, boxes, vt, vtb, region_dim=cfg.mm_region_hidden_size, vt_only=vt_only, **kw)[0]


def region_tokens(sd, cfg, feat):
    return projector(feat.bfloat16().float(), sd["proj"], "mm_projector_aux.", cfg.mm_projector_aux_type)    # :106-107


def llm_prefill(sd, cfg, ids, img_tok, reg_tok, gh, gw, return_all=False):
    """splice + rope index + LLM -> (embeds, pos, delta, final-norm hidden [L, d] (and all layer outputs), last-row logits [1, V])."""
    l = cfg.llm
    emb, nb, na = LO.splice(torch.tensor(ids), sd["llm"]["embed_tokens.weight"], img_tok, reg_tok)
    m = cfg.vit.spatial_merge_size
    pos, delta = LO.rope_index(nb, (gh // m, gw // m), na)
    kw = dict(n_layers=l.num_layers, n_heads=l.num_heads, n_kv=l.num_kv_heads, head_dim=l.head_dim, eps=l.rms_norm_eps, theta=l.rope_theta,
              sections=tuple(l.mrope_section))
    out = LO.llm_forward(sd["llm"], emb, pos, return_all=return_all, **kw)
    final, hs = out if return_all else (out, None)
    head = sd["llm"].get("lm_head.weight", sd["llm"]["embed_tokens.weight"])
    return dict(embeds=emb, pos=pos, delta=delta, final=final, hidden=hs, logits=final[-1:] @ head.t(), kw=kw)
