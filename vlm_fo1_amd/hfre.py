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
import os
from typing import List, Optional, Sequence, Union

import torch

from . import lib as _lib

FPN_STRIDES = (3.5, 7.0, 14.0, 28.0)  # reference :245


_env_applied = False


def _apply_env():
    """FO1_HFRE_UNROLL / FO1_HFRE_CHUNK / FO1_HFRE_BUDGET / FO1_HFRE_GRID tune the gather (A/B hooks, process-global, read once)."""
    global _env_applied
    if _env_applied:
        return
    _env_applied = True
    e = os.environ
    if _lib.ab_build() and any(k in e for k in ("FO1_HFRE_UNROLL", "FO1_HFRE_CHUNK", "FO1_HFRE_BUDGET", "FO1_HFRE_GRID")):
        _lib.check(_lib.load().fo1_hfre_set_tuning(int(e.get("FO1_HFRE_UNROLL", 8)), int(e.get("FO1_HFRE_CHUNK", 512)),
                                                   int(e.get("FO1_HFRE_BUDGET", 0)), int(e.get("FO1_HFRE_GRID", 0))), "fo1_hfre_set_tuning")


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
        # Built: aux + vt ('concat' / 'concat_aux_pos'), vt only (use_vt_region_feature_only), aux only (see below), with or without SimpleFPN,
        # 'bbox_based' / 'feature_map_based' / 'hybrid' position embedding (reference :327-335, :436-467), region LayerNorm (:365-372).
        # Not built (the engine refuses loudly): the 'mean*' / '*_sep_pos' fusions and per-region MLPs — with DaViT-L's 3840 aux
        # channels the reference itself cannot run them (mean adds a [N,3840] to a [N,2048|5120] tensor, *_sep_pos adds a 2880-wide
        # embedding, the MLPs are Linear(2048, .) on 3840 inputs: :373-432, :184-196).
        unsupported = []
        if region_feature_combination not in ("concat", "concat_aux_pos"):
            unsupported.append(f"region_feature_combination={region_feature_combination!r}")
        if use_separate_mlp_for_regions:
            unsupported.append("use_separate_mlp_for_regions")
        if pos_embedding_strategy not in ("bbox_based", "feature_map_based", "hybrid"):
            unsupported.append(f"pos_embedding_strategy={pos_embedding_strategy!r}")
        if use_vt_region_feature_only and not use_vision_tower_region_feature:
            unsupported.append("use_vt_region_feature_only without use_vision_tower_region_feature")
        if unsupported:
            raise NotImplementedError("HFRE variant not built for the MI355X engine: " + ", ".join(unsupported))
        # use_vision_tower_region_feature=False (the reference's DEFAULT, omchat_arch.py:23): the reference's own __call__ never binds
        # `out_box_feat` on that path (:368 is the only place the hybrid branch assigns it) and raises UnboundLocalError at :456 / :469
        # for every input — tests/test_oracle_hfre.py runs the reference in place to pin that.  Here the aux-only route computes what
        # the surrounding code evidently means: out = the aux block (:319-366), plus the box embedding from the AUX boxes normalised by
        # the aux map size / aux scale (the else-branch at :449-455), region LayerNorm = aux_region_norm only.  A labelled extension.
        self.use_vt_region_feature_only = use_vt_region_feature_only
        self.use_vision_tower_region_feature = use_vision_tower_region_feature
        self.region_feature_combination = region_feature_combination
        self.apply_region_layer_norm = apply_region_layer_norm
        self._ln = None          # (aux_w, aux_b, vt_w, vt_b) fp32 device tensors, set_region_norm()
        self.worklist = os.environ.get("FO1_HFRE_WORKLIST", "1") != "0"   # work-list kernels (0: the round-1 worst-case-grid form, for A/B)
        self.roi_output_size = roi_output_size
        self.region_feature_dim = region_feature_dim
        self.apply_position_embedding = apply_position_embedding
        self.pos_embedding_strategy = pos_embedding_strategy
        self._fm_pos = {}          # (H, W, C, batch, device) -> bf16 [batch*H*W, C] feature-map position table
        self.vision_tower_region_feature_dim = vision_tower_region_feature_dim
        self.vision_tower_spatial_scale = vision_tower_spatial_scale
        self.use_simpleFPN_for_vt = use_simpleFPN_for_vt
        self.aux_dims = tuple(aux_vision_tower_region_feature_dims)
        self.aux_vision_tower_spatial_scale = aux_vision_tower_spatial_scale
        self.simple_fpn = simple_fpn  # callable: [1,1280,gh,gw] -> 4 maps (engine op), when FPN is on
        self._ws = None

    def set_region_norm(self, aux_weight, aux_bias, vt_weight, vt_bias):
        """nn.LayerNorm parameters of `aux_region_norm` / `vt_region_norm` (reference :175-180); fp32 on the device."""
        def f(t):
            return None if t is None else t.detach().to(dtype=torch.float32).contiguous()
        self._ln = (f(aux_weight), f(aux_bias), f(vt_weight), f(vt_bias))

    # -- helpers ---------------------------------------------------------------
    def _feature_map_pos(self, H: int, W: int, C: int, batch: int, device) -> torch.Tensor:
        """generate_2d_position_embedding (reference :9-52) as token-major bf16 rows [batch*H*W, C]: y | x halves, each sin / cos
        interleaved over dim // 4 frequencies, coordinates normalised to [0, 1).  fp32 on the host with the reference's expression,
        cast like `pos_embed.to(feature.dtype)` (:208)."""
        key = (H, W, C, batch, str(device))
        t = self._fm_pos.get(key)
        if t is None:
            import math
            y = torch.arange(H, dtype=torch.float32) / H
            x = torch.arange(W, dtype=torch.float32) / W
            yg, xg = torch.meshgrid(y, x, indexing="ij")
            q = C // 4
            dim_t = torch.arange(q, dtype=torch.float32)
            dim_t = 10000 ** (2 * (dim_t // 2) / q) if q > 0 else torch.tensor([1.0])
            px = (xg.unsqueeze(-1) * (2 * math.pi)) / dim_t
            py = (yg.unsqueeze(-1) * (2 * math.pi)) / dim_t
            px = torch.stack((px.sin(), px.cos()), dim=-1).flatten(-2)
            py = torch.stack((py.sin(), py.cos()), dim=-1).flatten(-2)
            pe = torch.cat([py, px], dim=-1).reshape(H * W, -1)
            if pe.shape[1] != C:
                raise _lib.Fo1Error(f"feature-map position embedding: channel count {C} is not a multiple of 4")
            t = pe.to(torch.bfloat16).repeat(batch, 1).contiguous().to(device)
            if len(self._fm_pos) >= 64:
                self._fm_pos.pop(next(iter(self._fm_pos)))
            self._fm_pos[key] = t
        return t

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

    def pool(self, srcs: list, boxes: torch.Tensor, vt_boxes: Optional[torch.Tensor], vt_scale, pos_hw, out: Optional[torch.Tensor] = None,
             batch: int = 1, box_image: Optional[torch.Tensor] = None, img_strides: Optional[Sequence[int]] = None,
             ln_split: Optional[int] = None, out_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The C-ABI call.  srcs: HfreSource list (image 0 of every map); boxes fp32 [N,4] (all images' boxes, image of box n =
        box_image[n]); pos_hw = (pos_h, pos_w) normalisers of the box embedding.  Returns fp32 [1, N, region_feature_dim]."""
        L = _lib.load()
        dev = boxes.device
        N = boxes.shape[0]
        off = sum(s.C for s in srcs)
        if off != self.region_feature_dim:
            raise ValueError(f"feature channels {off} != region_feature_dim {self.region_feature_dim}")
        arr = (_lib.HfreSource * len(srcs))(*srcs)
        if out is None:
            out = torch.empty(1, N, off, dtype=torch.float32, device=dev)
        else:
            if out.dtype != torch.float32 or out.shape != (N, off) or out.stride(1) != 1 or out.device != dev:
                raise ValueError(f"out must be a device fp32 [{N}, {off}] row-major tensor")
            out = out.unsqueeze(0)
        sx, sy = (vt_scale if vt_scale is not None else (1.0, 1.0))
        pos_mode = 0
        if self.use_vt_region_feature_only:
            # reference :293-317 — the vt-only branch tests `apply_position_embedding` ALONE (whatever the strategy says, the vt box
            # embedding is added) and returns before any region LayerNorm
            pos_mode = 1 if self.apply_position_embedding else 0
        elif self.apply_position_embedding and self.pos_embedding_strategy in ("bbox_based", "hybrid"):     # reference :438-440
            pos_mode = 2 if (self.region_feature_combination == "concat_aux_pos" or not self.use_vision_tower_region_feature) else 1
        from . import ops as _ops
        use_ln = self.apply_region_layer_norm and not self.use_vt_region_feature_only
        if use_ln and self._ln is None:
            raise _lib.Fo1Error("apply_region_layer_norm: call set_region_norm() with the checkpoint's LayerNorm parameters first")
        if out_bf16 is not None and (out_bf16.dtype != torch.bfloat16 or out_bf16.shape != (N, off) or out_bf16.stride(1) != 1 or out_bf16.device != dev):
            raise ValueError(f"out_bf16 must be a device bf16 [{N}, {off}] row-major tensor")
        if self.worklist or use_ln or batch > 1 or out_bf16 is not None:
            _apply_env()
            need = L.fo1_hfre_ex_workspace_bytes(arr, len(srcs), N)
            ws = _ops._workspace("hfre_ex", dev, max(int(need), 1))
            opts = _lib.HfreOpts()
            opts.batch = batch
            opts.box_image = box_image.data_ptr() if box_image is not None else None
            for i in range(8):
                opts.img_stride[i] = int(img_strides[i]) if (img_strides is not None and i < len(srcs)) else 0
            opts.ln_on = 1 if use_ln else 0
            if use_ln:
                aw, ab, vw, vb = self._ln
                opts.ln_split = int(ln_split if ln_split is not None else 0)
                opts.ln_w0, opts.ln_b0 = (aw.data_ptr(), ab.data_ptr()) if aw is not None else (None, None)
                opts.ln_w1, opts.ln_b1 = (vw.data_ptr(), vb.data_ptr()) if vw is not None else (None, None)
                opts.ln_eps = 1e-5
            if out_bf16 is not None:
                opts.out_bf16, opts.out_bf16_ld = out_bf16.data_ptr(), out_bf16.stride(0)
            rc = L.fo1_hfre_region_pool_ex(arr, len(srcs), boxes.data_ptr(), N, vt_boxes.data_ptr() if vt_boxes is not None else None,
                                              float(sx), float(sy), self.roi_output_size, pos_mode, float(pos_hw[1]), float(pos_hw[0]),
                                              out.data_ptr(), out.stride(1), off, ctypes.byref(opts), ws.data_ptr(), ws.numel(),
                                              _lib.current_stream_ptr())
            _lib.check(rc, "fo1_hfre_region_pool_ex")
            self._ws = ws
            return out
        need = L.fo1_hfre_workspace_bytes(arr, len(srcs), N)
        # scratch from the owner-scoped pool (vlm_fo1_amd/ops.py): never resized in place — a captured graph may hold the pointer
        self._ws = _ops._workspace("hfre", dev, max(int(need), 1))
        rc = L.fo1_hfre_region_pool(arr, len(srcs), boxes.data_ptr(), N, vt_boxes.data_ptr() if vt_boxes is not None else None,
                                    float(sx), float(sy), self.roi_output_size, pos_mode, float(pos_hw[1]), float(pos_hw[0]),
                                    out.data_ptr(), out.stride(1), off, self._ws.data_ptr(), self._ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "fo1_hfre_region_pool")
        return out

    def __call__(self, aux_multi_level_features: List[torch.Tensor], aux_boxes: Union[torch.Tensor, List[torch.Tensor]],
                 vt_multi_level_features=None, vt_boxes: Union[torch.Tensor, List[torch.Tensor], None] = None,
                 vt_scale=None, out: Optional[torch.Tensor] = None, batch: int = 1, box_image: Optional[torch.Tensor] = None,
                 out_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns fp32 [1, N, region_feature_dim] like the reference (:469).  `vt_boxes` may be omitted when `vt_scale=(sx, sy)` is
        given (vt = aux * scale in-kernel).  `out`: optional fp32 [N, C_region] row-contiguous destination.  batch > 1: every map
        tensor holds `batch` same-size images stacked ([1,C,H,W] views of image 0 whose storage continues image by image, H*W*ld
        elements apart), `aux_boxes` holds all images' boxes and `box_image` (device int32 [N]) the image of each.  `out_bf16`: optional
        bf16 [N, C_region] destination written by the same finish kernel (the `.to(tower dtype)` of omchat_qwen2_5_vl.py:106)."""
        boxes = aux_boxes[0] if isinstance(aux_boxes, (list, tuple)) else aux_boxes
        dev = aux_multi_level_features[0].device if aux_multi_level_features else boxes.device
        if dev.type != "cuda":
            raise _lib.Fo1Error("HFRE runs on the HIP device only (got %s)" % dev)
        boxes = boxes.to(device=dev, dtype=torch.float32).contiguous()
        vtb = None
        if vt_boxes is not None:
            vtb = vt_boxes[0] if isinstance(vt_boxes, (list, tuple)) else vt_boxes
            vtb = vtb.to(device=dev, dtype=torch.float32).contiguous()
        elif vt_scale is None and self.use_vision_tower_region_feature:
            raise ValueError("need vt_boxes or vt_scale")
        keep: list = []
        srcs, strides = [], []
        off = 0
        H0 = W0 = 0
        if not self.use_vt_region_feature_only:
            H0 = max(f.shape[2] for f in aux_multi_level_features)
            W0 = max(f.shape[3] for f in aux_multi_level_features)
            fm_pos = self.apply_position_embedding and self.pos_embedding_strategy in ("feature_map_based", "hybrid")
            for f in aux_multi_level_features:
                if fm_pos:
                    # reference :327-335 / :198-211: feature + pos_embed.to(feature.dtype), a bf16 add on every aux level BEFORE the
                    # fp32 upsample / roi_align (so it is not linear in the pooled result: the sum is rounded to bf16 first)
                    from . import ops as _ops
                    f = _token_major_bf16(f)
                    _, C, H, W = f.shape
                    rows = f.permute(0, 2, 3, 1)
                    if not rows.is_contiguous():
                        rows = rows.contiguous()
                    if batch > 1 and rows.untyped_storage().nbytes() < (rows.storage_offset() + batch * H * W * C) * 2:
                        raise _lib.Fo1Error("feature-map position embedding: the stacked maps of a batched call must be contiguous")
                    rows = rows.as_strided((batch * H * W, C), (C, 1)) if batch > 1 else rows.reshape(H * W, C)
                    summed = _ops.add(rows, self._feature_map_pos(H, W, C, batch, rows.device))
                    keep.append(summed)
                    f = summed[:H * W].view(1, H, W, C).permute(0, 3, 1, 2)        # image-0 view; storage continues image by image
                srcs.append(self._src(f, (H0, W0), self.aux_vision_tower_spatial_scale, 0, off, keep))
                strides.append(f.shape[2] * f.shape[3] * srcs[-1].ld)
                off += f.shape[1]
        aux_channels = off
        gh = gw = 0
        if self.use_vision_tower_region_feature:
            if self.use_simpleFPN_for_vt:
                vt_in = vt_multi_level_features  # [1,1280,gh,gw]
                gh, gw = vt_in.shape[-2:]
                fpn_maps = self.simple_fpn(vt_in)
                for f, st in zip(fpn_maps, FPN_STRIDES):
                    srcs.append(self._src(f, f.shape[2:], 1.0 / st, 1, off, keep))
                    strides.append(f.shape[2] * f.shape[3] * srcs[-1].ld)
                    off += f.shape[1]
            else:
                gh = max(f.shape[-2] for f in vt_multi_level_features)
                gw = max(f.shape[-1] for f in vt_multi_level_features)
                for f in vt_multi_level_features:
                    srcs.append(self._src(f, f.shape[2:], self.vision_tower_spatial_scale, 1, off, keep))
                    strides.append(f.shape[2] * f.shape[3] * srcs[-1].ld)
                    off += f.shape[1]
        # reference :446-455 — box normalisers: vt map size / vt scale, or (aux-based embedding) aux map size / aux scale
        if self.use_vision_tower_region_feature and self.region_feature_combination != "concat_aux_pos":
            pos_hw = (gh / self.vision_tower_spatial_scale, gw / self.vision_tower_spatial_scale)
        else:
            pos_hw = (H0 / self.aux_vision_tower_spatial_scale, W0 / self.aux_vision_tower_spatial_scale)
        return self.pool(srcs, boxes, vtb, vt_scale, pos_hw, out=out, batch=batch, box_image=box_image, img_strides=strides,
                         ln_split=aux_channels, out_bf16=out_bf16)

    forward = __call__
