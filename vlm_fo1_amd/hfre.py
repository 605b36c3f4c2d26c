"""Host-side mirror of the reference's HFREModule for the MI355X engine.

Same constructor keywords and call signature as
`vlm_fo1/model/multimodal_visual_prompt_encoder/hybrid_finegrained_region_encoder.py:106-292`
(HFREModule), but the arithmetic is one call into libfo1hip.so
(`fo1_hfre_region_pool`, include/fo1.h).  Supported configuration = the product
one (SURVEY §8a "config keys"): aux pyramid + vision-tower features, 'concat'
fusion, 'bbox_based' box position embedding, with or without SimpleFPN on the vt
branch.  Anything else raises NotImplementedError — loudly, no fallback.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Union

import torch

from . import lib as _lib

FPN_STRIDES = (3.5, 7.0, 14.0, 28.0)  # reference :245


def _token_major_bf16(x: torch.Tensor) -> torch.Tensor:
    """[1,C,H,W] (any strides/dtype) -> a bf16 tensor whose memory is [H*W, C].
    The reference's towers already hand out NCHW *views* of token-major memory
    (DaViT: modeling_davit.py:493; ViT: qwen2_5_vl_encoder.py:76-79), so this is
    normally free."""
    assert x.dim() == 4 and x.shape[0] == 1, f"expected [1,C,H,W], got {tuple(x.shape)}"
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x


class HFREModule:
    def __init__(self, roi_output_size: int = 7, region_feature_dim: int = 1024,
                 apply_position_embedding: bool = False, pos_embedding_strategy: str = "bbox_based",
                 use_vt_region_feature_only: bool = False, use_vision_tower_region_feature: bool = False,
                 region_feature_combination: str = "concat", use_separate_mlp_for_regions: bool = False,
                 apply_region_layer_norm: bool = False, vision_tower_region_feature_dim: int = 5120,
                 vision_tower_spatial_scale: float = 1 / 14, use_simpleFPN_for_vt: bool = False,
                 aux_vision_tower_region_feature_dims: Sequence[int] = (256, 512, 1024, 2048),
                 aux_vision_tower_spatial_scale: float = 0.25, simple_fpn=None):
        unsupported = []
        if use_vt_region_feature_only:
            unsupported.append("use_vt_region_feature_only")
        if not use_vision_tower_region_feature:
            unsupported.append("use_vision_tower_region_feature=False")
        if region_feature_combination != "concat":
            unsupported.append(f"region_feature_combination={region_feature_combination!r}")
        if use_separate_mlp_for_regions:
            unsupported.append("use_separate_mlp_for_regions")
        if apply_region_layer_norm:
            unsupported.append("apply_region_layer_norm")
        if apply_position_embedding and pos_embedding_strategy != "bbox_based":
            unsupported.append(f"pos_embedding_strategy={pos_embedding_strategy!r}")
        if unsupported:
            raise NotImplementedError("HFRE variant not built for the MI355X engine: " + ", ".join(unsupported))
        self.roi_output_size = roi_output_size
        self.region_feature_dim = region_feature_dim
        self.apply_position_embedding = apply_position_embedding
        self.vision_tower_region_feature_dim = vision_tower_region_feature_dim
        self.vision_tower_spatial_scale = vision_tower_spatial_scale
        self.use_simpleFPN_for_vt = use_simpleFPN_for_vt
        self.aux_dims = tuple(aux_vision_tower_region_feature_dims)
        self.aux_vision_tower_spatial_scale = aux_vision_tower_spatial_scale
        self.simple_fpn = simple_fpn  # callable: [1,1280,gh,gw] -> 4 maps (engine op), when FPN is on
        self._ws = None

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _src(x: torch.Tensor, roi_hw, scale: float, box_space: int, out_off: int, keep: list):
        x = _token_major_bf16(x)
        _, C, H, W = x.shape
        tm = x.permute(0, 2, 3, 1)  # [1,H,W,C]
        if tm.stride(3) != 1 or tm.stride(2) < C or tm.stride(1) != W * tm.stride(2):
            tm = tm.contiguous()
        keep.append(tm)
        ld = tm.stride(2)
        return _lib.HfreSource(tm.data_ptr(), H, W, C, ld, roi_hw[0], roi_hw[1], float(scale), box_space, out_off)

    def __call__(self, aux_multi_level_features: List[torch.Tensor], aux_boxes: Union[torch.Tensor, List[torch.Tensor]],
                 vt_multi_level_features=None, vt_boxes: Union[torch.Tensor, List[torch.Tensor], None] = None,
                 vt_scale=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns fp32 [1, N, region_feature_dim] like the reference (:469).  `vt_boxes`
        may be omitted when `vt_scale=(sx, sy)` is given (vt = aux * scale in-kernel).  `out`: optional fp32 [N, C_region]
        row-contiguous destination (the batched engine hands out row slices of one buffer)."""
        L = _lib.load()
        boxes = aux_boxes[0] if isinstance(aux_boxes, (list, tuple)) else aux_boxes
        dev = aux_multi_level_features[0].device
        if dev.type != "cuda":
            raise _lib.Fo1Error("HFRE runs on the HIP device only (got %s)" % dev)
        boxes = boxes.to(device=dev, dtype=torch.float32).contiguous()
        N = boxes.shape[0]
        vtb = None
        if vt_boxes is not None:
            vtb = vt_boxes[0] if isinstance(vt_boxes, (list, tuple)) else vt_boxes
            vtb = vtb.to(device=dev, dtype=torch.float32).contiguous()
        elif vt_scale is None:
            raise ValueError("need vt_boxes or vt_scale")
        keep: list = []
        srcs = []
        H0 = max(f.shape[2] for f in aux_multi_level_features)
        W0 = max(f.shape[3] for f in aux_multi_level_features)
        off = 0
        for f in aux_multi_level_features:
            srcs.append(self._src(f, (H0, W0), self.aux_vision_tower_spatial_scale, 0, off, keep))
            off += f.shape[1]
        if self.use_simpleFPN_for_vt:
            vt_in = vt_multi_level_features  # [1,1280,gh,gw]
            gh, gw = vt_in.shape[-2:]
            fpn_maps = self.simple_fpn(vt_in)
            for f, s in zip(fpn_maps, FPN_STRIDES):
                srcs.append(self._src(f, f.shape[2:], 1.0 / s, 1, off, keep))
                off += f.shape[1]
        else:
            gh = max(f.shape[-2] for f in vt_multi_level_features)
            gw = max(f.shape[-1] for f in vt_multi_level_features)
            for f in vt_multi_level_features:
                srcs.append(self._src(f, f.shape[2:], self.vision_tower_spatial_scale, 1, off, keep))
                off += f.shape[1]
        if off != self.region_feature_dim:
            raise ValueError(f"feature channels {off} != region_feature_dim {self.region_feature_dim}")
        arr = (_lib.HfreSource * len(srcs))(*srcs)
        need = L.fo1_hfre_workspace_bytes(arr, len(srcs), N)
        # scratch from the owner-scoped pool (vlm_fo1_amd/ops.py): never resized in place — a captured graph may hold the pointer
        from . import ops as _ops
        self._ws = _ops._workspace("hfre", dev, max(int(need), 1))
        if out is None:
            out = torch.empty(1, N, off, dtype=torch.float32, device=dev)
        else:
            if out.dtype != torch.float32 or out.shape != (N, off) or out.stride(1) != 1 or out.device != dev:
                raise ValueError(f"out must be a device fp32 [{N}, {off}] row-major tensor")
            out = out.unsqueeze(0)
        # reference :446-448 — image size = vt map size / vt spatial scale (python floats)
        pos_w = gw / self.vision_tower_spatial_scale
        pos_h = gh / self.vision_tower_spatial_scale
        sx, sy = (vt_scale if vt_scale is not None else (1.0, 1.0))
        rc = L.fo1_hfre_region_pool(arr, len(srcs), boxes.data_ptr(), N,
                                    vtb.data_ptr() if vtb is not None else None, float(sx), float(sy),
                                    self.roi_output_size, 1 if self.apply_position_embedding else 0,
                                    float(pos_w), float(pos_h), out.data_ptr(), out.stride(1), off,
                                    self._ws.data_ptr(), self._ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "fo1_hfre_region_pool")
        return out

    forward = __call__
