"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement of the Hybrid Fine-grained Region Encoder exactly as the
reference computes it (direct algorithm: upsample → concat → roi_align →
spatial mean → fuse → sine box embedding), NOT the separable reformulation
the HIP kernel uses.  Follows

  vlm_fo1/model/multimodal_visual_prompt_encoder/hybrid_finegrained_region_encoder.py
    gen_sineembed_for_position            :55-103
    HFREModule.extract_vt_region_feature  :230-273
    HFREModule.__call__                   :275-469
  vlm_fo1/model/language_model/omchat_qwen2_5_vl.py  encode_regions :75-128
  torchvision==0.21.0 ops.roi_align (not vendored; restated here and in
  oracle/roi_align_ref.c)

Pinning: `load_reference_hfre()` imports the reference's own HFREModule from
/root/reference (when present) with our roi_align injected for the missing
torchvision symbol; tests/test_oracle_hfre.py checks this restatement against
it and against the committed golden vectors (tests/golden/hfre_*.npz, made by
tests/golden/make_hfre_golden.py).  roi_align itself has no golden vector in
the reference tree → "parity unpinned" for that one function.
"""
from __future__ import annotations

import ctypes
import importlib.util
import math
import os
import sys
import types
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


# --------------------------------------------------------------------------
# roi_align — independent torch restatement (vectorised per box)
# --------------------------------------------------------------------------
def roi_align_torch(inp: torch.Tensor, boxes, output_size: int, spatial_scale: float = 1.0,
                    sampling_ratio: int = -1, aligned: bool = False) -> torch.Tensor:
    """inp [1,C,H,W] (any strides) float32/float64; boxes Tensor[N,4] or [Tensor[N,4]].
    Returns [N,C,P,P].  Coordinates are computed in inp.dtype like torchvision's
    CPU kernel (template type T)."""
    if isinstance(boxes, (list, tuple)):
        assert len(boxes) == 1
        boxes = boxes[0]
    assert inp.dim() == 4 and inp.shape[0] == 1
    dt = inp.dtype
    _, C, H, W = inp.shape
    P = output_size
    feat = inp[0].permute(1, 2, 0).reshape(H * W, C)  # [HW, C]
    out = torch.zeros(boxes.shape[0], C, P, P, dtype=dt)
    off = 0.5 if aligned else 0.0
    ss = torch.tensor(spatial_scale, dtype=dt)
    for n in range(boxes.shape[0]):
        b = boxes[n].to(dt)
        x1 = b[0] * ss - off
        y1 = b[1] * ss - off
        x2 = b[2] * ss - off
        y2 = b[3] * ss - off
        rw = x2 - x1
        rh = y2 - y1
        if not aligned:
            rw = torch.clamp(rw, min=1.0)
            rh = torch.clamp(rh, min=1.0)
        bh = rh / P
        bw = rw / P
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh / P)))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw / P)))
        count = max(gh * gw, 1)
        ph = torch.arange(P, dtype=dt)[:, None]
        iy = torch.arange(gh, dtype=dt)[None, :]
        ix = torch.arange(gw, dtype=dt)[None, :]
        ys = (y1 + ph * bh + (iy + 0.5) * bh / gh).reshape(-1)  # [P*gh]
        xs = (x1 + ph * bw + (ix + 0.5) * bw / gw).reshape(-1)  # [P*gw]

        def axis(v, size):
            valid = ~((v < -1.0) | (v > size))
            v = torch.clamp(v, min=0.0)
            lo = v.to(torch.int64)
            top = lo >= size - 1
            lo = torch.where(top, torch.full_like(lo, size - 1), lo)
            hi = torch.where(top, lo, lo + 1)
            v = torch.where(top, lo.to(dt), v)
            l = v - lo.to(dt)
            h = 1.0 - l
            return valid, lo, hi, h, l

        vy, ylo, yhi, hy, ly = axis(ys, H)
        vx, xlo, xhi, hx, lx = axis(xs, W)
        valid = (vy[:, None] & vx[None, :]).to(dt)  # [Py, Px]

        def gather(yi, xi):
            idx = (yi[:, None] * W + xi[None, :]).reshape(-1)
            return feat[idx].reshape(yi.numel(), xi.numel(), C)

        val = (gather(ylo, xlo) * (hy[:, None] * hx[None, :])[..., None]
               + gather(ylo, xhi) * (hy[:, None] * lx[None, :])[..., None]
               + gather(yhi, xlo) * (ly[:, None] * hx[None, :])[..., None]
               + gather(yhi, xhi) * (ly[:, None] * lx[None, :])[..., None])
        val = val * valid[..., None]
        val = val.reshape(P, gh, P, gw, C).sum(dim=(1, 3)) / count  # [P,P,C]
        out[n] = val.permute(2, 0, 1)
    return out


# --------------------------------------------------------------------------
# roi_align — C restatement (oracle/roi_align_ref.c) via ctypes
# --------------------------------------------------------------------------
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        from oracle.build import build
        _LIB = ctypes.CDLL(build())
        i64, f32, i32, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_void_p
        for name in ("oracle_roi_align_f32", "oracle_roi_align_mean_f32"):
            fn = getattr(_LIB, name)
            fn.argtypes = [vp, i64, i64, i64, i64, i64, i64, vp, i64, f32, i32, i32, i32, vp]
            fn.restype = None
    return _LIB


def roi_align_c(inp: torch.Tensor, boxes, output_size: int, spatial_scale: float = 1.0,
                sampling_ratio: int = -1, aligned: bool = False, mean: bool = False) -> torch.Tensor:
    """Same signature as torchvision.ops.roi_align (fp32 only).  Honors strides, so
    channels-last views are read in place."""
    if isinstance(boxes, (list, tuple)):
        assert len(boxes) == 1
        boxes = boxes[0]
    assert inp.dtype == torch.float32 and inp.dim() == 4 and inp.shape[0] == 1
    _, C, H, W = inp.shape
    _, sc, sh, sw = inp.stride()
    boxes = boxes.detach().to(torch.float32).contiguous()
    N = boxes.shape[0]
    P = output_size
    if mean:
        out = torch.empty(N, C, dtype=torch.float32)
        fn = _lib().oracle_roi_align_mean_f32
    else:
        out = torch.empty(N, C, P, P, dtype=torch.float32)
        fn = _lib().oracle_roi_align_f32
    fn(inp.data_ptr(), C, H, W, sc, sh, sw, boxes.data_ptr(), N,
       float(spatial_scale), P, sampling_ratio, int(aligned), out.data_ptr())
    return out


# --------------------------------------------------------------------------
# sine box embedding  (hybrid_finegrained_region_encoder.py:55-103)
# --------------------------------------------------------------------------
def sine_embed(pos: torch.Tensor, d: int) -> torch.Tensor:
    """pos [B,N,4] = (cx, cy, w, h) normalised; returns [B,N,4d] ordered (y, x, w, h)."""
    scale = 2 * math.pi
    dim_t = torch.arange(d, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / d)

    def enc(v):
        p = (v * scale)[:, :, None] / dim_t
        return torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2)

    return torch.cat((enc(pos[:, :, 1]), enc(pos[:, :, 0]), enc(pos[:, :, 2]), enc(pos[:, :, 3])), dim=2)


def box_pos_embed(boxes: torch.Tensor, img_w: float, img_h: float, d: int) -> torch.Tensor:
    """xyxy px → normalised cxcywh → sine embedding (reference :456-466, fp32)."""
    b = boxes.clone().to(torch.float32)
    b[:, [0, 2]] = b[:, [0, 2]] / img_w
    b[:, [1, 3]] = b[:, [1, 3]] / img_h
    b[:, 2] = b[:, 2] - b[:, 0]
    b[:, 3] = b[:, 3] - b[:, 1]
    b[:, 0] = b[:, 0] + b[:, 2] / 2
    b[:, 1] = b[:, 1] + b[:, 3] / 2
    return sine_embed(b.unsqueeze(0), d)


# --------------------------------------------------------------------------
# HFRE — literal restatement of HFREModule.__call__ for the supported variants
#   concat (+/- SimpleFPN on the vt branch), bbox_based position embedding.
# --------------------------------------------------------------------------
def feature_map_pos_embed(H: int, W: int, C: int) -> torch.Tensor:
    """generate_2d_position_embedding (reference :11-52): [H, W, C] fp32 — y half then x half, each sin / cos interleaved over
    C // 4 frequencies, coordinates normalised to [0, 1)."""
    y = torch.arange(H, dtype=torch.float32) / H
    x = torch.arange(W, dtype=torch.float32) / W
    yg, xg = torch.meshgrid(y, x, indexing="ij")
    q = C // 4
    dim_t = torch.arange(q, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / q) if q > 0 else torch.tensor([1.0])
    px = xg.unsqueeze(-1) * (2 * math.pi) / dim_t
    py = yg.unsqueeze(-1) * (2 * math.pi) / dim_t
    px = torch.stack((px.sin(), px.cos()), dim=-1).flatten(-2)
    py = torch.stack((py.sin(), py.cos()), dim=-1).flatten(-2)
    return torch.cat([py, px], dim=-1)


def hfre_oracle(aux_maps: Sequence[torch.Tensor], aux_boxes: torch.Tensor,
                vt_maps: Optional[Sequence[torch.Tensor]], vt_boxes: Optional[torch.Tensor],
                *, region_dim: int, grid_hw, vt_strides: Optional[Sequence[float]] = None,
                vt_spatial_scale: float = 1 / 14, aux_spatial_scale: float = 0.25,
                roi_size: int = 7, apply_pos: bool = True, roi_align=None, region_ln: Optional[dict] = None,
                pos_from: str = "vt", vt_only: bool = False, aux_only: bool = False, strategy: str = "bbox_based") -> torch.Tensor:
    """Supported product configuration (reference :319-363, :368-383, :436-467 with
    use_vision_tower_region_feature=True, combination='concat', strategy 'bbox_based').

    aux_maps: 4x[1,C_l,H_l,W_l] (any dtype/strides).
    vt_maps : the 4 captured ViT maps [1,1280,gh,gw] (vt_strides=None: concatenated,
              one roi_align at 1/14) or the 4 SimpleFPN outputs (vt_strides=[3.5,7,14,28]:
              one roi_align per level, reference :245-257).
    grid_hw : (gh, gw) of the ViT patch grid; the position embedding normalises the vt
              boxes by grid*14 (reference :443-448 — under FPN `vt_multi_level_features`
              is the single grid-resolution input map, so the same value).
    Returns fp32 [1,N,region_dim]."""
    # Variants beyond the default configuration: region_ln = dict(aux_w, aux_b, vt_w, vt_b) applies the reference's fp32
    # nn.LayerNorm(eps 1e-5) to the aux and the vt block before fusion (:365-372); pos_from='aux' is 'concat_aux_pos' (box
    # embedding from the aux boxes, normalised by the aux map size / aux scale, :443-455); vt_only is use_vt_region_feature_only
    # (:293-317: vt block + vt box embedding, no aux tower).
    # aux_only is use_vision_tower_region_feature=False — NOT a restatement: the reference raises UnboundLocalError there (`out_box_feat`
    # is never bound, :456/:469; tests/golden/hfre_variants.npz `aux_only_error`).  It states the engine's labelled extension: the
    # reference's aux block (:319-366, pinned through the `nopos` golden) + the box embedding of the else-branch at :449-455 (aux
    # boxes / aux map size / aux scale) with region_dim = 3840.
    ra = roi_align or roi_align_c
    aux_boxes = aux_boxes.float()
    if aux_only:
        assert not vt_only
        vt_maps, vt_boxes, pos_from = None, aux_boxes, "aux"
    vt_boxes = vt_boxes.float()
    H0 = max(f.shape[2] for f in aux_maps)
    W0 = max(f.shape[3] for f in aux_maps)
    aux = None
    # strategy 'feature_map_based' / 'hybrid' (:327-335, :206-227): `feature + pos_embed.to(feature.dtype)` on every aux level in the
    # maps' own dtype (bf16 on the product path) BEFORE the fp32 upsample; 'feature_map_based' then skips the box embedding (:438-440).
    # The vt-only branch ignores the strategy (:293-317).
    fm = strategy in ("feature_map_based", "hybrid") and apply_pos and not vt_only
    box_pos = apply_pos and (vt_only or strategy in ("bbox_based", "hybrid"))
    if not vt_only:
        cat = []
        for lvl, f in enumerate(aux_maps):
            if fm:
                _, C, H, W = f.shape
                f = f + feature_map_pos_embed(H, W, C).permute(2, 0, 1).unsqueeze(0).to(f.dtype)
            f = f.float()
            if lvl != 0:
                f = F.interpolate(f, size=(H0, W0), mode="bilinear", align_corners=False)
            cat.append(f)
        cat = torch.cat(cat, dim=1)
        aux = ra(cat, [aux_boxes], output_size=roi_size, spatial_scale=aux_spatial_scale)
        aux = aux.mean(dim=(2, 3)).reshape(1, aux.shape[0], aux.shape[1])
        if region_ln is not None:
            aux = F.layer_norm(aux, (aux.shape[-1],), region_ln["aux_w"].float(), region_ln["aux_b"].float(), 1e-5)
    if aux_only:
        vt = None
    elif vt_strides is not None:
        per = []
        for f, s in zip(vt_maps, vt_strides):
            r = ra(f.float(), [vt_boxes], output_size=roi_size, spatial_scale=1.0 / s)
            per.append(r.mean(dim=(2, 3)))
        vt = torch.cat(per, dim=1).unsqueeze(0)
    else:
        vcat = torch.cat(list(vt_maps), dim=1).float()
        r = ra(vcat, [vt_boxes], output_size=roi_size, spatial_scale=vt_spatial_scale)
        vt = r.mean(dim=(2, 3)).reshape(1, r.shape[0], r.shape[1])
    if region_ln is not None and not vt_only and not aux_only:
        vt = F.layer_norm(vt, (vt.shape[-1],), region_ln["vt_w"].float(), region_ln["vt_b"].float(), 1e-5)
    out = vt if vt_only else (aux if aux_only else torch.cat([aux, vt], dim=-1))
    assert out.shape[-1] == region_dim, (out.shape, region_dim)
    if box_pos:
        gh, gw = grid_hw
        if pos_from == "aux" and not vt_only:
            out = out + box_pos_embed(aux_boxes, W0 / aux_spatial_scale, H0 / aux_spatial_scale, region_dim // 4)
        else:
            out = out + box_pos_embed(vt_boxes, gw / vt_spatial_scale, gh / vt_spatial_scale, region_dim // 4)
    return out


# --------------------------------------------------------------------------
# The reference's own modules, imported from /root/reference (this container only)
# --------------------------------------------------------------------------
def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vlm_fo1"))


def _load_by_path(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_hfre(roi_align=None):
    """Returns (HFREModule, SimpleFP, gen_sineembed_for_position) classes from the
    reference tree with `torchvision.ops.roi_align` replaced by our restatement
    (torchvision is not installed).  The reference files are executed in place —
    nothing is copied."""
    assert reference_available(), "/root/reference not present (GPU box?)"
    ra = roi_align or roi_align_c
    base = os.path.join(REFERENCE_ROOT, "vlm_fo1", "model", "multimodal_visual_prompt_encoder")
    saved = {k: sys.modules.get(k) for k in list(sys.modules)
             if k == "torchvision" or k.startswith("torchvision.") or k == "vlm_fo1" or k.startswith("vlm_fo1.")}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        tv = types.ModuleType("torchvision")
        tvo = types.ModuleType("torchvision.ops")
        tvo.roi_align = ra
        tv.ops = tvo
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = tvo
        for pkg in ("vlm_fo1", "vlm_fo1.model", "vlm_fo1.model.multimodal_visual_prompt_encoder"):
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        fpn = _load_by_path("vlm_fo1.model.multimodal_visual_prompt_encoder.simple_fpn",
                            os.path.join(base, "simple_fpn.py"))
        hf = _load_by_path("vlm_fo1.model.multimodal_visual_prompt_encoder.hybrid_finegrained_region_encoder",
                           os.path.join(base, "hybrid_finegrained_region_encoder.py"))
        return hf.HFREModule, fpn.SimpleFP, hf.gen_sineembed_for_position
    finally:
        for k in [k for k in sys.modules if k == "torchvision" or k.startswith("torchvision.")
                  or k == "vlm_fo1" or k.startswith("vlm_fo1.")]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
