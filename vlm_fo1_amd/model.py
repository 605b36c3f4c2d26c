"""The VLM-FO1 hot path as one engine object:

    pixel_values, aux image, boxes, prompt ids
        -> Qwen2.5-VL ViT  ──► image tokens ──► mm_projector ─────────────┐
        │        └ last full-attention map ─► SimpleFPN ─┐                  │
        -> DaViT-L ─► 4 pyramid maps ────────────────────┴► HFRE ─► mm_projector_aux ─► region tokens
        -> splice (embed_tokens + image + region tokens) -> mRoPE ids -> LLM prefill -> greedy decode

Mirrors `OmChatQwen25VLForCausalLM.encode_images / encode_regions /
prepare_inputs_labels_for_qwen2_5_vl_multimodal / forward` (omchat_qwen2_5_vl.py:44-128,135-463,466-532)
and `OmChatMetaModel.__init__` (omchat_arch.py:8-33).  Host code orchestrates libfo1hip.so kernels only."""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .davit import DAVIT_LARGE, DaViT
from .fpn import SimpleFPN
from .hfre import HFREModule
from .llm import DEFAULT_REGION_INDEX, IMAGE_TOKEN_INDEX, LLMConfig, QwenLLM
from .vit import QwenViT, ViTConfig


@dataclass
class FO1Config:
    vit: ViTConfig = field(default_factory=ViTConfig)
    llm: LLMConfig = field(default_factory=LLMConfig)
    mm_projector_type: str = "mlp2x_gelu"
    mm_projector_aux_type: str = "mlp2x_gelu"
    mm_use_simpleFPN_for_vt: bool = True
    mm_region_hidden_size: int = 5888           # 3840 (aux pyramid) + 4 x 512 (FPN); 8960 without FPN
    mm_roi_output_size: int = 7
    mm_apply_position_embedding: bool = True
    mm_pos_embedding_strategy: str = "bbox_based"      # 'bbox_based' | 'feature_map_based' | 'hybrid' (omchat_arch.py:21)
    mm_apply_region_layer_norm: bool = False      # HFRE :365-372: nn.LayerNorm on the aux and the vt block before the box embedding
    mm_region_feature_combination: str = "concat"  # 'concat' | 'concat_aux_pos'
    mm_use_vt_region_feature_only: bool = False
    # False = region features from the aux tower only (the reference's default value, omchat_arch.py:23 — a configuration the
    # reference itself cannot run: its HFRE raises UnboundLocalError, see vlm_fo1_amd/hfre.py; built here as a labelled extension)
    mm_use_vision_tower_region_feature: bool = True


class Projector:
    """`build_vision_projector(_aux)` (multimodal_projector/builder.py:39-115): identity / linear / mlpNx_gelu."""

    def __init__(self, kind: str, state: Dict[str, torch.Tensor], prefix: str, device):
        self.kind = kind
        self.layers: List[Tuple[torch.Tensor, torch.Tensor]] = []

        def dv(t):
            return t.to(device=device, dtype=torch.bfloat16).contiguous()

        if kind == "identity":
            return
        if kind == "linear":
            self.layers.append((dv(state[prefix + "weight"]), dv(state[prefix + "bias"])))
            return
        m = re.match(r"^mlp(\d+)x_gelu$", kind)
        if not m:
            raise NotImplementedError(f"projector type {kind!r} is not built for the MI355X engine")
        for i in range(int(m.group(1))):
            self.layers.append((dv(state[f"{prefix}{2 * i}.weight"]), dv(state[f"{prefix}{2 * i}.bias"])))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from . import stage_abi
        if stage_abi.enabled():      # the same launches, sequenced by fo1_projector_forward (csrc/stages.hip)
            return stage_abi.projector_forward(self.layers, x)
        n = len(self.layers)
        for i, (w, b) in enumerate(self.layers):
            x = ops.gemm(x, w, b, act=ops.ACT_GELU if i + 1 < n else ops.ACT_NONE)
        return x


class FO1Engine:
    def __init__(self, cfg: FO1Config, weights: Dict[str, Dict[str, torch.Tensor]], device="cuda"):
        """weights: sub state-dicts keyed 'vit', 'davit', 'fpn', 'llm', 'proj' (mm_projector.* / mm_projector_aux.*),
        each with the checkpoint's key names (prefixes stripped)."""
        self.cfg = cfg
        self.dev = torch.device(device)
        self.vit = QwenViT(cfg.vit, weights["vit"], device)
        self.davit = DaViT(weights["davit"], device)
        self.use_vt = bool(cfg.mm_use_vision_tower_region_feature)
        if cfg.mm_use_vt_region_feature_only and not self.use_vt:
            raise NotImplementedError("mm_use_vt_region_feature_only needs mm_use_vision_tower_region_feature")
        self.fpn = SimpleFPN(weights["fpn"], device) if (cfg.mm_use_simpleFPN_for_vt and self.use_vt) else None
        # which ViT hidden states the region branch reads: none (aux-only), the last full-attention map (FPN) or all four (:82-85)
        self.capture = "none" if not self.use_vt else ("last" if self.fpn is not None else "all")
        self.llm = QwenLLM(cfg.llm, weights["llm"], device, lm_head=weights["llm"].get("lm_head.weight"))
        self.mm_projector = Projector(cfg.mm_projector_type, weights["proj"], "mm_projector.", device)
        self.mm_projector_aux = Projector(cfg.mm_projector_aux_type, weights["proj"], "mm_projector_aux.", device)
        self.hfre = HFREModule(roi_output_size=cfg.mm_roi_output_size, region_feature_dim=cfg.mm_region_hidden_size,
                               apply_position_embedding=cfg.mm_apply_position_embedding, pos_embedding_strategy=cfg.mm_pos_embedding_strategy,
                               use_vision_tower_region_feature=self.use_vt, region_feature_combination=cfg.mm_region_feature_combination,
                               use_vt_region_feature_only=cfg.mm_use_vt_region_feature_only,
                               apply_region_layer_norm=cfg.mm_apply_region_layer_norm,
                               vision_tower_region_feature_dim=2048 if cfg.mm_use_simpleFPN_for_vt else 4 * cfg.vit.hidden_size,
                               vision_tower_spatial_scale=1 / cfg.vit.patch_size,
                               use_simpleFPN_for_vt=cfg.mm_use_simpleFPN_for_vt, aux_vision_tower_spatial_scale=0.25)
        if cfg.mm_apply_region_layer_norm:
            pw = weights["proj"]
            g = lambda k: pw[k].to(self.dev) if k in pw else None
            self.hfre.set_region_norm(g("aux_region_norm.weight"), g("aux_region_norm.bias"), g("vt_region_norm.weight"), g("vt_region_norm.bias"))
        self._dummy_box = torch.tensor([[0., 10., 0., 10.]], device=self.dev)  # omchat_qwen2_5_vl.py:90-91
        import collections
        self._graphs = collections.OrderedDict()   # signature -> captured prefill graph (LRU, GRAPH_CACHE entries)
        self._seen = {}                            # signature -> sightings before capture
        # Two-stream tower overlap (DaViT || ViT+FPN) is OFF: measured on MI355X / ROCm 7.2 a forked hipGraph replays at
        # 39.7 ms vs 21.9 ms single-stream (cross-stream joins serialise the node launches), see profiles/README.md.
        self._ws_owner = self.llm._ws_owner = ops.new_owner(self)   # scratch buffers are keyed by this token (ops.workspace_scope)
        self.stage_hook = None   # callable(stage_name) at stage boundaries; measurement only, never set while capturing a graph

    # ---- encoders ------------------------------------------------------------------------------
    def replica(self) -> "FO1Engine":
        """An engine that shares every weight tensor with this one but owns its per-request state (KV cache, decode state,
        captured graphs, HFRE workspace).  One replica per HIP stream lets independent images overlap on the GPU: most of a
        batch-1 pass's kernels under-fill 256 CUs, so two passes in flight give ~1.35x the images/s of one (profiles/README.md)."""
        import copy
        r = copy.copy(self)
        r.llm = self.llm.replica()
        r.hfre = copy.copy(self.hfre)
        r._ws_owner = r.llm._ws_owner = ops.new_owner(r)   # scratch buffers are keyed by this token (ops.workspace_scope)
        import collections
        r._graphs = collections.OrderedDict()
        r._seen = {}
        r._dec = None
        r._dec_extra = []
        r._pool_svc = getattr(self, "_pool_svc", None)      # ONE decode pool per GPU: replicas feed the same service
        r._dec_stream0 = None
        r.stage_hook = None
        return r

    # ---- fp8 linears (BASELINE configs[4]) -------------------------------------------------------
    FP8_PRESETS = {"all": ("vit.wqkv", "vit.wgu", "vit.wd", "llm.wqkv", "llm.wgu", "llm.wdown"),
                   "mlp": ("vit.wgu", "vit.wd", "llm.wgu", "llm.wdown"),        # attention projections stay bf16
                   "llm-mlp": ("llm.wgu", "llm.wdown")}
    FP8_DEFAULT = FP8_PRESETS["all"]

    def enable_fp8(self, which: Sequence[str] = FP8_DEFAULT) -> int:
        """W8A8 e4m3 for the named projections of every ViT block / LLM layer in the packed pass (products with >= ops.FP8_MIN_ROWS
        rows; decode and small products keep the bf16 weights): weights are quantised per output channel once, activations per
        token in front of each product.  Not the reference's numerics (it has no fp8 path): DESIGN.md section 10 holds the
        measured deviation table.  Captured graphs are dropped.  Returns the number of weights registered."""
        n = 0
        if isinstance(which, str):
            which = self.FP8_PRESETS[which]
        for name in which:
            part, key = name.split(".")
            layers = self.vit.blocks if part == "vit" else self.llm.layers
            for w in layers:
                k = ops.register_fp8_weight(w[key])
                if k is not None:
                    self._fp8_keys = getattr(self, "_fp8_keys", []) + [k]
                    n += 1
        self._graphs.clear()
        self._seen.clear()
        return n

    def disable_fp8(self) -> None:
        """Back to bf16 for the weights THIS engine registered (replicas share weights and therefore the routing)."""
        ops.clear_fp8_weights(getattr(self, "_fp8_keys", []))
        self._fp8_keys = []
        self._graphs.clear()
        self._seen.clear()

    def _mark(self, stage: str):
        """Stage boundary for measurement (bench.py sets `stage_hook` in its eager profiling pass only)."""
        if self.stage_hook is not None:
            self.stage_hook(stage)

    def encode_images(self, pixel_values: torch.Tensor, gh: int, gw: int):
        """-> (image tokens [S/4, d_llm], captured ViT maps (token-major raster))  (encode_images :44-72)."""
        tokens, feats = self.vit.forward(pixel_values, gh, gw, capture=self.capture)
        self._mark("qwen_vit+merger")
        out = self.mm_projector(tokens)
        self._mark("mm_projector")
        return out, feats

    def encode_regions(self, aux_image: torch.Tensor, boxes: Optional[torch.Tensor], vt_feats: List[torch.Tensor], gh: int, gw: int):
        """-> region tokens [N, d_llm]  (encode_regions :75-108).  boxes: fp32 [N,4] xyxy in aux-image pixels."""
        # vt-only region features never read the aux pyramid (HFRE :293-317): the DaViT pass is skipped, the result is the same
        aux_maps, aux_sizes = ([], []) if self.cfg.mm_use_vt_region_feature_only else self.davit.forward(aux_image)
        self._mark("davit_large")
        if boxes is None or boxes.shape[0] == 0:
            boxes = self._dummy_box
        boxes = boxes.to(device=self.dev, dtype=torch.float32)
        H, W = aux_image.shape[-2:]
        p = self.cfg.vit.patch_size
        # reference :94-99 — python-float scales, one fp32 multiply per coordinate
        sh, sw = (gh * p) / H, (gw * p) / W

        def nchw(t, hw):  # token-major [H*W, C] -> the NCHW *view* the reference hands to HFRE (no copy)
            return t.view(1, hw[0], hw[1], t.shape[1]).permute(0, 3, 1, 2)

        aux_views = [nchw(t, s) for t, s in zip(aux_maps, aux_sizes)]
        if not self.use_vt:
            vt_in = None                                # aux-only region features (reference :109-126)
        elif self.fpn is not None:
            fpn_maps, fpn_sizes = self.fpn.forward(vt_feats[-1], gh, gw)
            self._mark("simple_fpn")
            fpn_views = [nchw(t, s) for t, s in zip(fpn_maps, fpn_sizes)]
            self.hfre.simple_fpn = lambda x: fpn_views
            vt_in = nchw(vt_feats[-1], (gh, gw))
        else:
            vt_in = [nchw(t, (gh, gw)) for t in vt_feats]
        feat16 = torch.empty(boxes.shape[0], self.cfg.mm_region_hidden_size, dtype=torch.bfloat16, device=self.dev)
        self.hfre(aux_views, [boxes], vt_in, None, vt_scale=(sw, sh), out_bf16=feat16)     # fp32 [N, C_region] + its bf16 cast (:106), one kernel
        self._mark("hfre_region_pool")
        out = self.mm_projector_aux(feat16)                                                # :107
        self._mark("mm_projector_aux")
        return out

    # ---- a batch of images: everything up to the first generated token of each ---------------------------------
    def _regions_batch(self, auxs, aux_stack, boxes, want, vt_last, bp, grids, img_of, boxes_cat=None, box_image=None):
        """encode_regions (:75-108) for every request that has regions -> (region tokens [sum N, d_llm] or None, per-request row
        ranges).  `auxs` / `grids` / `bp` describe the UNIQUE images of the pass, `img_of[r]` the image of request r (several prompts
        may share one image: BASELINE configs[4]'s 300 proposals are 3 prompts of 100 over one image — the towers run once per
        image).  DaViT / SimpleFPN run ONCE over all images: stacked when they share the aux size and the patch grid, packed row-wise
        with per-image geometry tables when they do not (forward_ragged).  The HFRE gather is one launch for every box of every
        image in the stacked case, one call per request on views of its image's rows otherwise; it writes straight into the
        request's rows of one [sum N, C_region] fp32 buffer."""
        idx = [i for i, w in enumerate(want) if w]
        if not idx:
            return None, [(0, 0)] * len(want)
        p = self.cfg.vit.patch_size
        feat = torch.empty(sum(boxes[i].shape[0] for i in idx), self.cfg.mm_region_hidden_size, dtype=torch.float32, device=self.dev)
        feat16 = torch.empty(feat.shape, dtype=torch.bfloat16, device=self.dev)   # the `.to(tower dtype)` of :106, written by the HFRE finish kernel
        imgs = sorted({img_of[i] for i in idx})                 # unique images that feed the region branch
        all_imgs = len(imgs) == len(auxs)
        uniform = aux_stack is not None and all_imgs and len(set(grids)) == 1 and len(idx) == len(want)
        vt_only = self.cfg.mm_use_vt_region_feature_only

        def nchw(t, hw, r0=0):  # rows [r0, r0 + h*w) of a token-major map -> the NCHW *view* the reference hands to HFRE
            return t[r0:r0 + hw[0] * hw[1]].view(1, hw[0], hw[1], t.shape[1]).permute(0, 3, 1, 2)

        def scales(u):          # reference :94-99 — python-float scales, one fp32 multiply per coordinate
            gh, gw = grids[u]
            H, W = auxs[u].shape[-2:]
            return (gw * p) / W, (gh * p) / H

        def ranges_in_order():
            row, rg = 0, {}
            for i in idx:
                rg[i] = (row, row + boxes[i].shape[0])
                row += boxes[i].shape[0]
            return rg

        ranges = ranges_in_order()

        def hfre_request(i, aux_views, fpn_views, vt_views):
            """One request's boxes on views of its image's maps -> its rows of `feat`."""
            u = img_of[i]
            if not self.use_vt:
                vt_in = None
            elif self.fpn is not None:
                self.hfre.simple_fpn = lambda x, v=fpn_views: v
                vt_in = vt_views
            else:
                vt_in = vt_views
            r0, r1 = ranges[i]
            self.hfre(aux_views, [boxes[i]], vt_in, None, vt_scale=scales(u), out=feat[r0:r1], out_bf16=feat16[r0:r1])

        if uniform:
            G = len(auxs)
            aux_maps, aux_sizes = ([], []) if vt_only else self.davit.forward(aux_stack)
            self._mark("davit_large")
            gh, gw = grids[0]
            n = gh * gw
            if self.fpn is not None:
                fpn_maps, fpn_sizes = self.fpn.forward(vt_last, gh, gw, batch=G)
                self._mark("simple_fpn")
            if G > 1 and boxes_cat is not None and box_image is not None:
                # one launch for every box of every request: views of image 0, the kernel steps image by image through the stacks
                aux_views = [nchw(t, s) for t, s in zip(aux_maps, aux_sizes)]
                if not self.use_vt:
                    vt_in = None
                elif self.fpn is not None:
                    fv = [nchw(t, s) for t, s in zip(fpn_maps, fpn_sizes)]
                    self.hfre.simple_fpn = lambda x, v=fv: v
                    vt_in = nchw(vt_last, (gh, gw))
                else:
                    vt_in = [nchw(t, (gh, gw)) for t in vt_last]
                self.hfre(aux_views, [boxes_cat], vt_in, None, vt_scale=scales(0), out=feat, batch=G, box_image=box_image, out_bf16=feat16)
            else:
                for i in idx:
                    u = img_of[i]
                    aux_views = [nchw(t, s, u * s[0] * s[1]) for t, s in zip(aux_maps, aux_sizes)]
                    fv = [nchw(t, s, u * s[0] * s[1]) for t, s in zip(fpn_maps, fpn_sizes)] if self.fpn is not None else None
                    vv = None if not self.use_vt else (nchw(vt_last, (gh, gw), u * n) if self.fpn is not None else [nchw(t, (gh, gw), u * n) for t in vt_last])
                    hfre_request(i, aux_views, fv, vv)
            self._mark("hfre_region_pool")
        elif len(imgs) > 1 and self.RAGGED_TOWERS:
            # images of different sizes: DaViT / SimpleFPN still run ONCE over all of them (rows packed image by image, the spatial
            # kernels read per-image geometry tables: davit.forward_ragged / fpn.forward_ragged)
            aux_maps, aplan = ([], None) if vt_only else self.davit.forward_ragged([auxs[u] for u in imgs])
            self._mark("davit_large")
            fpn_maps, fplan = [], None
            if self.fpn is not None:
                if all_imgs:
                    vt_sel, r0s = vt_last, [bp.row0[u] for u in imgs]
                else:       # only the images that have regions: their raster maps, packed back to back
                    parts = [vt_last[bp.row0[u]:bp.row0[u] + grids[u][0] * grids[u][1]] for u in imgs]
                    vt_sel, r0s, o = torch.cat(parts, 0), [], 0
                    for t in parts:
                        r0s.append(o)
                        o += t.shape[0]
                fpn_maps, fplan = self.fpn.forward_ragged(vt_sel, [grids[u] for u in imgs], r0s)
                self._mark("simple_fpn")
            # the plans own the device geometry tables (ops.ImgSegs) the launches above read: a captured graph of this pass holds
            # their raw pointers, so the pass result keeps the plans alive past the towers' 64-entry plan caches
            self._pass_keep += [aplan, fplan]
            slot = {u: j for j, u in enumerate(imgs)}
            for i in idx:
                u = img_of[i]
                j = slot[u]
                gh, gw = grids[u]
                aux_views = [nchw(t, aplan.sizes[l][j], aplan.row0[l][j]) for l, t in enumerate(aux_maps)]
                fv = [nchw(t, fplan.sizes[l][j], fplan.row0[l][j]) for l, t in enumerate(fpn_maps)] if self.fpn is not None else None
                r0 = bp.row0[u]
                vv = None if not self.use_vt else (nchw(vt_last, (gh, gw), r0) if self.fpn is not None else [nchw(t, (gh, gw), r0) for t in vt_last])
                hfre_request(i, aux_views, fv, vv)
            self._mark("hfre_region_pool")
        else:
            for u in imgs:       # image by image (one image with regions, or RAGGED_TOWERS off: the round-2 path, kept for A/B)
                aux_maps, aux_sizes = ([], []) if vt_only else self.davit.forward(auxs[u].unsqueeze(0))
                self._mark("davit_large")
                gh, gw = grids[u]
                r0 = bp.row0[u]
                fv = None
                if self.fpn is not None:
                    fpn_maps, fpn_sizes = self.fpn.forward(vt_last[r0:r0 + gh * gw], gh, gw, batch=1)
                    self._mark("simple_fpn")
                    fv = [nchw(t, s) for t, s in zip(fpn_maps, fpn_sizes)]
                aux_views = [nchw(t, s) for t, s in zip(aux_maps, aux_sizes)]
                vv = None if not self.use_vt else (nchw(vt_last, (gh, gw), r0) if self.fpn is not None else [nchw(t, (gh, gw), r0) for t in vt_last])
                for i in idx:
                    if img_of[i] == u:
                        hfre_request(i, aux_views, fv, vv)
            self._mark("hfre_region_pool")
        out = self.mm_projector_aux(feat16)                                                # :107
        self._mark("mm_projector_aux")
        return out, [ranges.get(i, (0, 0)) for i in range(len(want))]

    def _device_batch(self, st, meta):
        keep: list = []                      # host objects whose device tables this pass's launches (and a graph of them) point at
        with ops.workspace_scope(self._ws_owner), ops.keep_scope(keep):
            self._pass_keep = keep
            grids = meta["grids"]
            tokens, feats, bp = self.vit.forward_batch(st["pix"], grids, capture=self.capture)
            keep.append(bp)                  # (the ViT's batch plans are evicted from a 64-entry cache the same way)
            self._mark("qwen_vit+merger")
            image_tokens = self.mm_projector(tokens)
            self._mark("mm_projector")
            vt_last = None if not self.use_vt else (feats[-1] if self.fpn is not None else feats)
            region_tokens, ranges = self._regions_batch(st["aux"], st.get("aux_stack"), st["boxes"], meta["want"], vt_last, bp, grids,
                                                        meta["img_of"], st.get("boxes_cat"), st.get("box_image"))
            emb = self.llm.embed_rows(st["plan"], image_tokens, region_tokens)
            self._mark("splice")
            last, logits, toks = self.llm.prefill_packed(emb, st["cos"], st["sin"], meta["seqs"], st["last"])
            self._mark("llm_prefill+lm_head+argmax")
            return dict(image_tokens=image_tokens, region_tokens=region_tokens, embeds=emb, last_hidden=last, logits=logits,
                        next_tokens=toks, region_ranges=ranges, row0=bp.row0, _keep=keep)

    PREFILL_MAX = 32       # requests per packed prefill pass of generate_batch
    PREFILL_ROWS = 65536   # ... and ViT patch rows per pass: 32 COCO-sized images are 50k rows; a group of the datasets' largest images (32 x 10 800
                           # patches: CountBench / Pixmo through evaluation/eval_countbench.py) would otherwise be ONE pass whose SimpleFPN scratch
                           # alone is 44 GB (round 5: the driver-level CountBench run hit it)
    PREFILL_AUX_PIXELS = 24 << 20   # ... and aux-image pixels per pass: the DaViT / SimpleFPN scratch follows the AUX size, which `dynamic` mode
                           # takes from the native resolution, not from max_pixels — small-grid images with large aux tensors must not share one oversized pass

    def split_passes(self, requests: Sequence[dict]) -> List[List[dict]]:
        """Consecutive requests -> packed prefill passes of <= PREFILL_MAX requests, <= PREFILL_ROWS ViT patch rows and <= PREFILL_AUX_PIXELS aux pixels (a single request
        larger than the row budget is its own pass).  Order is kept: the caller's results stay in request order."""
        out, cur, rows, apix = [], [], 0, 0
        aux_budget = getattr(self, "PREFILL_AUX_PIXELS", None)

        def aux_pixels(r):
            a = r.get("aux")
            return int(a.shape[-2]) * int(a.shape[-1]) if a is not None and hasattr(a, "shape") else 0
        for r in requests:
            new_image = r.get("image_id") is None or not any(q.get("image_id") == r["image_id"] for q in cur)
            n = int(r["grid"][0]) * int(r["grid"][1]) if new_image else 0
            a = aux_pixels(r) if new_image else 0
            if cur and (len(cur) >= self.PREFILL_MAX or rows + n > self.PREFILL_ROWS or (aux_budget is not None and apix + a > aux_budget)):
                out.append(cur)
                cur, rows, apix = [], 0, 0
                n, a = int(r["grid"][0]) * int(r["grid"][1]), aux_pixels(r)
            cur.append(r)
            rows += n
            apix += a
        if cur:
            out.append(cur)
        return out
    DECODE_CONCURRENT = True   # decode groups of one pass advance together on their own streams (False: one after the other; A/B)
    DECODE_GROUPS = 1      # minimum number of decode groups when a pass has more sequences than one group holds
    DECODE_MAX_GROUP = 32  # sequences per decode group (<= BatchDecoder.MAX_BATCH); 16 = round 2's one-MFMA-column-group decode (A/B)
    RAGGED_TOWERS = True   # images of different sizes share one DaViT / SimpleFPN pass (False: image by image, the round-2 path; A/B)
    SHARE_PREFIX = True    # prompts over ONE image (`image_id`) run their common prefix rows (system text + image tokens) through the LLM once
                           # (llm.plan_batch(share_prefix=True), fo1_attention_prefix_bf16); False: every prompt's full rows (round 3; A/B)
    GRAPH_CACHE = 8        # captured prefill graphs kept per engine (LRU); each holds its own activation pool
    CAPTURE_AFTER = 1      # sightings of a signature before it is captured: one-off shapes (a dataset of ragged images) run eagerly

    def prefill_batch(self, requests: Sequence[dict], use_graph: bool = False) -> List[dict]:
        """requests: dicts with ids (sentinel id list), pix [S,1176], grid (gh, gw), aux [3,H,W], boxes [N,4] or None.
        All images go through ONE packed pass of every stage (the batch-aware splice of omchat_qwen2_5_vl.py:380-416, done
        varlen: no padding to the longest prompt).  Returns one dict per request: image_tokens, region_tokens, embeds,
        last_hidden [1,d], logits [1,V], next_token, position_ids, rope_delta.  use_graph=True replays a captured hipGraph of
        the pass for this shape signature once the signature has been seen CAPTURE_AFTER times (LRU of GRAPH_CACHE graphs)."""
        m = self.cfg.vit.spatial_merge_size
        B = len(requests)
        # unique images of the pass: requests that carry the same `image_id` (any hashable) share ONE image — its towers run once
        # and every such prompt's <image> block addresses the same rows of the image-token table (configs[4]: 300 proposals =
        # 3 prompts of 100 over one image; the reference would run the whole model three times)
        img_of, first_of, seen_ids = [], [], {}
        for i, r in enumerate(requests):
            iid = r.get("image_id")
            if iid is None or iid not in seen_ids:
                if iid is not None:
                    seen_ids[iid] = len(first_of)
                img_of.append(len(first_of))
                first_of.append(i)
            else:
                u = seen_ids[iid]
                f = requests[first_of[u]]
                if tuple(f["grid"]) != tuple(r["grid"]) or f["pix"].shape != r["pix"].shape or f["aux"].shape != r["aux"].shape:
                    raise ValueError(f"requests with image_id {iid!r} describe different images")
                img_of.append(u)
        U = len(first_of)
        grids, n_img, n_reg, want, boxes, prompts = [], [], [], [], [], []
        for r in requests:
            gh, gw = r["grid"]
            b = r.get("boxes")
            w = (b is not None) or any(t == DEFAULT_REGION_INDEX for t in r["ids"])
            if b is None or b.shape[0] == 0:
                b = self._dummy_box
            grids.append((int(gh), int(gw))); n_img.append((gh // m) * (gw // m)); want.append(bool(w))
            boxes.append(b.to(device=self.dev, dtype=torch.float32)); n_reg.append(b.shape[0] if w else 0); prompts.append(r["ids"])
        ugrids = [grids[i] for i in first_of]
        tok_base, o = [], 0
        for i in first_of:
            tok_base.append(o)
            o += n_img[i]
        hp = self.llm.plan_batch(prompts, n_img, n_reg, [(g[0] // m, g[1] // m) for g in grids], img_base=[tok_base[u] for u in img_of],
                                 share_prefix=self.SHARE_PREFIX and U < B)
        if self.llm.reserve(hp["rows"]):
            # the caches moved (cache_epoch bumped): every captured pass holds dead pointers and is unreachable by key — release the
            # graphs and their private activation pools now instead of waiting for 8 new captures to evict them
            self._graphs.clear()
            self._seen.clear()
        meta = dict(grids=tuple(ugrids), want=tuple(want), seqs=tuple(hp["seqs"]), img_of=tuple(img_of))
        ureqs = [requests[i] for i in first_of]
        pix = ureqs[0]["pix"] if U == 1 else torch.cat([r["pix"].to(self.dev) for r in ureqs], 0)
        auxs = [r["aux"] if r["aux"].dim() == 3 else r["aux"][0] for r in ureqs]
        host = dict(plan=hp["plan"], cos=hp["cos"], sin=hp["sin"], last=hp["last"],
                    box_image=torch.tensor([img_of[i] for i, b in enumerate(boxes) if want[i] for _ in range(b.shape[0])] or [0], dtype=torch.int32))
        key = (meta["grids"], tuple(tuple(a.shape) for a in auxs), tuple(n_reg), meta["seqs"], meta["want"], meta["img_of"], pix.dtype, auxs[0].dtype,
               self.llm.cache_epoch)
        ent = self._graphs.get(key) if use_graph else None
        if use_graph and ent is None:
            seen = self._seen.get(key, 0)
            if seen >= self.CAPTURE_AFTER:
                ent = self._capture(key, pix, auxs, boxes, host, meta)
            else:
                if len(self._seen) >= 4096:
                    self._seen.pop(next(iter(self._seen)))
                self._seen[key] = seen + 1
        if ent is None:
            st = dict(pix=pix, aux=auxs, boxes=boxes, **{k: v.to(self.dev) for k, v in host.items()})
            if len({tuple(a.shape) for a in auxs}) == 1:      # same-size aux images: one DaViT pass over the stack (input staging)
                st["aux_stack"] = auxs[0].unsqueeze(0) if U == 1 else torch.stack([a.to(self.dev) for a in auxs], 0)
                if U > 1 and all(want):
                    st["boxes_cat"] = torch.cat(boxes, 0)
            res = self._device_batch(st, meta)
        else:
            self._graphs.move_to_end(key)
            g, st, res, keep = ent
            off = 0
            for r in ureqs:                         # device -> device slices; host tensors upload blocking
                n = r["pix"].shape[0]
                st["pix"][off:off + n].copy_(r["pix"], non_blocking=r["pix"].is_cuda)
                off += n
            for d, a in zip(st["aux"], auxs):
                d.copy_(a, non_blocking=a.is_cuda)
            for d, b in zip(st["boxes"], boxes):
                d.copy_(b, non_blocking=True)
            # Host-built inputs: async uploads from pageable tensors.  The runtime stages the bytes at call time (measured: an
            # event-guarded pinned buffer or a blocking copy serialises graph launch and execution); the last few source
            # tensors are kept alive anyway so their memory cannot be re-used under a pending copy.
            for k, src in host.items():
                st[k].copy_(src, non_blocking=True)
            keep.append(host)
            if len(keep) > 8:
                del keep[0]
            with ops.graph_lock.replay():
                g.replay()
        outs = []
        for i, (o, L, Lp, *pre) in enumerate(hp["seqs"]):
            r0, r1 = res["region_ranges"][i]
            i0 = res["row0"][img_of[i]] // (m * m)
            if not pre:
                emb = res["embeds"][o:o + L]
            elif ent is None:     # a prompt whose prefix rows are shared: its embedding rows are two pieces (assembled for introspection only,
                emb = torch.cat([res["embeds"][pre[0]:pre[0] + pre[1]], res["embeds"][o:o + L - pre[1]]], 0)     # never under graph replay)
            else:
                emb = None
            outs.append(dict(image_tokens=res["image_tokens"][i0:i0 + n_img[i]],
                             region_tokens=res["region_tokens"][r0:r1] if res["region_tokens"] is not None and want[i] else None,
                             embeds=emb, last_hidden=res["last_hidden"][i:i + 1], logits=res["logits"][i:i + 1],
                             next_token=res["next_tokens"][i:i + 1], position_ids=hp["pos"][i], rope_delta=hp["delta"][i],
                             cache_rows=(o, L)))
        self._last_batch = hp
        self._last_next_tokens = res["next_tokens"]
        if B == 1:                                   # the single request sits at cache position 0: decode can continue in place
            self.llm.kv_len = hp["seqs"][0][1]
            self.llm.rope_delta = hp["delta"][0]
        return outs

    def generate_batch(self, requests: Sequence[dict], max_new_tokens: int = 512, stop_ids: Sequence[int] = (), use_graph: bool = True) -> List[List[int]]:
        """Greedy generation for a batch of requests: one packed prefill, then the batched decode loop (vlm_fo1_amd.llm.BatchDecoder):
        weights streamed once per step for all sequences, stop rule and bookkeeping on the device.  Returns the new ids per request
        (stop token included, like HF generate)."""
        from .llm import BatchDecoder, run_decoders
        out: List[List[int]] = []
        if getattr(self, "_pool_svc", None) is not None:
            # decode pool: every pass's sequences join the shared pool; this call's later passes prefill while its earlier ones decode
            handles = [self.submit_batch(grp, max_new_tokens, stop_ids, use_graph) for grp in self.split_passes(requests)]
            for h in handles:
                out += h.result()
            return out
        for grp in self.split_passes(requests):
            self.prefill_batch(grp, use_graph=use_graph)          # ONE packed pass for the whole group (its GEMMs see every image's rows)
            hp = self._last_batch
            first = self._last_next_tokens
            n = len(grp)
            gmax = min(BatchDecoder.MAX_BATCH, self.DECODE_MAX_GROUP)
            if n <= gmax:
                dec = self._decoder()
                dec.start(hp["seqs"], hp["delta"], first[:n], max_new_tokens, stop_ids)
                out += dec.run(max_new_tokens, use_graph=use_graph)
                continue
            # more sequences than one decode group carries (32 = two 16-column MFMA groups per weight fragment; DECODE_MAX_GROUP = 16
            # restores round 2's groups for A/B): balanced groups (25 -> 13 + 12), each relocated out of the
            # prefill cache into its own decoder's slots and advanced on its own stream, all groups together (llm.run_decoders)
            k = max(-(-n // gmax), min(self.DECODE_GROUPS, n))
            cuts = [round(j * n / k) for j in range(k + 1)]
            if not self.DECODE_CONCURRENT:      # A/B: the groups one after the other on the caller's stream
                dec = self._decoder()
                for j in range(k):
                    a, b = cuts[j], cuts[j + 1]
                    dec.start(hp["seqs"][a:b], hp["delta"][a:b], first[a:b], max_new_tokens, stop_ids)
                    out += dec.run(max_new_tokens, use_graph=use_graph)
                continue
            decs, streams = self._decoders(k)
            cur = torch.cuda.current_stream()
            for j in range(k):
                a, b = cuts[j], cuts[j + 1]
                streams[j].wait_stream(cur)                       # the prefill (and its first tokens) are on the caller's stream
                with torch.cuda.stream(streams[j]):
                    decs[j].start(hp["seqs"][a:b], hp["delta"][a:b], first[a:b], max_new_tokens, stop_ids)
            for ids in run_decoders(decs[:k], streams[:k], max_new_tokens, use_graph=use_graph):
                out += ids
            for j in range(k):
                cur.wait_stream(streams[j])                       # the next pass must not overwrite the prefill cache under a relocate
        return out

    # ---- continuous batching: one decode pool per GPU, shared by every replica (vlm_fo1_amd/serving.py) -------------------------------
    DECODE_POOLS = 1       # decode pools per GPU stepping concurrently on their own streams (serving.PoolGroup); FO1_DECODE_POOLS overrides

    def enable_decode_pool(self, slots: int = 128, slot_rows: int = 1024, steps_per_round: int = 4, pools: Optional[int] = None):
        """From now on generate_batch() / submit_batch() of this engine AND of the replicas made from it afterwards hand their sequences
        to the GPU's decode pool(s) (llm.DecodePool): the sequences of successive prefill passes — of any replica — share every decode step
        (64 / 128 per weight stream instead of <= 32 per pass).  pools > 1: that many pools advance concurrently on their own HIP streams
        (serving.PoolGroup).  Returns the service."""
        from .serving import PoolGroup, PoolService
        if getattr(self, "_pool_svc", None) is None:
            n = int(pools if pools is not None else os.environ.get("FO1_DECODE_POOLS", self.DECODE_POOLS))
            if n > 1:
                self._pool_svc = PoolGroup(self.llm, pools=n, slots=slots, slot_rows=slot_rows, steps_per_round=steps_per_round)
            else:
                self._pool_svc = PoolService(self.llm, slots=slots, slot_rows=slot_rows, steps_per_round=steps_per_round)
        return self._pool_svc

    def disable_decode_pool(self) -> None:
        svc = getattr(self, "_pool_svc", None)
        if svc is not None:
            svc.close()
        self._pool_svc = None

    def submit_batch(self, requests: Sequence[dict], max_new_tokens: int = 512, stop_ids: Sequence[int] = (), use_graph: bool = True):
        """One packed prefill pass (<= PREFILL_MAX requests), then its sequences join the decode pool.  Returns a serving.PoolHandle as
        soon as the pool has taken the K / V^T rows over: the caller may start its next pass while these sequences decode;
        handle.result() -> the new ids per request (stop token included, like HF generate)."""
        svc = getattr(self, "_pool_svc", None)
        if svc is None:
            raise RuntimeError("submit_batch needs enable_decode_pool()")
        if len(requests) > self.PREFILL_MAX:
            raise ValueError(f"submit_batch takes one prefill pass (<= {self.PREFILL_MAX} requests)")
        self.prefill_batch(requests, use_graph=use_graph)
        hp = self._last_batch
        h = svc.submit(self.llm, hp["seqs"], hp["delta"], self._last_next_tokens[:len(requests)], max_new_tokens, stop_ids)
        h.wait_relocated()
        return h

    def _decoders(self, k: int):
        """k decode groups' decoders (the first is the engine's own) and the side streams they run on."""
        from .llm import BatchDecoder
        extra = self.__dict__.setdefault("_dec_extra", [])
        while len(extra) < k - 1:
            extra.append((BatchDecoder(self.llm), torch.cuda.Stream(device=self.dev)))
        if not hasattr(self, "_dec_stream0") or self._dec_stream0 is None:
            self._dec_stream0 = torch.cuda.Stream(device=self.dev)
        decs = [self._decoder()] + [d for d, _ in extra[:k - 1]]
        streams = [self._dec_stream0] + [s for _, s in extra[:k - 1]]
        for d in decs:
            if d.llm is not self.llm:
                raise RuntimeError("decoder bound to another engine")
        return decs, streams

    def _decoder(self):
        from .llm import BatchDecoder
        if getattr(self, "_dec", None) is None or self._dec.llm is not self.llm:
            self._dec = BatchDecoder(self.llm)
        return self._dec

    def _capture(self, key, pix, auxs, boxes, host, meta):
        # Captures always run with inference mode OFF: the CUDA generator's graph bookkeeping tensors are created by the first live
        # capture and updated in place by later ones — if the first ran under the caller's torch.inference_mode() (the reference's
        # inference.py:46) a later capture outside it fails ("Inplace update to inference tensor outside InferenceMode").
        with ops.graph_lock.capture(), torch.inference_mode(False):   # exclusive: no other thread captures or launches meanwhile
            # static input buffers must be ordinary tensors even when the caller runs under torch.inference_mode()
            # (the reference's inference.py:46 does): they are updated in place later
            with torch.inference_mode(False):
                st = dict(pix=pix.clone(), boxes=[b.clone() for b in boxes], **{k: v.clone().to(self.dev) for k, v in host.items()})
                if len({tuple(a.shape) for a in auxs}) == 1:
                    st["aux_stack"] = torch.stack([a.to(self.dev) for a in auxs], 0)
                    st["aux"] = list(st["aux_stack"].unbind(0))       # views: refreshing them refreshes the stack
                    if len(auxs) > 1 and all(meta["want"]):
                        st["boxes_cat"] = torch.cat(st["boxes"], 0)
                        st["boxes"] = list(st["boxes_cat"].split([b.shape[0] for b in boxes], 0))
                else:
                    st["aux"] = [a.clone() for a in auxs]
            s = torch.cuda.Stream()   # warm-up on a side stream (allocates every lazily-created scratch buffer), then capture
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._device_batch(st, meta)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL watchdog threads may touch the runtime meanwhile
                res = self._device_batch(st, meta)
            ent = (g, st, res, [])
            self._graphs[key] = ent
            while len(self._graphs) > self.GRAPH_CACHE:      # LRU: the evicted graph's private pool is released with it
                self._graphs.popitem(last=False)
            return ent

    def prefill(self, input_ids: Sequence[int], pixel_values: torch.Tensor, grid_hw: Tuple[int, int], aux_image: torch.Tensor,
                boxes: Optional[torch.Tensor], use_graph: bool = False):
        """One image = a batch of one (same kernels, same packed layout)."""
        out = self.prefill_batch([dict(ids=input_ids, pix=pixel_values, grid=grid_hw, aux=aux_image, boxes=boxes)], use_graph)[0]
        if boxes is None and not any(t == DEFAULT_REGION_INDEX for t in input_ids):
            out["region_tokens"] = None
        return out

    def generate(self, input_ids: Sequence[int], pixel_values, grid_hw, aux_image, boxes, max_new_tokens: int = 512,
                 stop_ids: Sequence[int] = (), use_graph: bool = False) -> List[int]:
        """Greedy decode (do_sample=False in every reference caller: mm_utils.py:640-654)."""
        out = self.prefill(input_ids, pixel_values, grid_hw, aux_image, boxes, use_graph=use_graph)
        self.llm.reserve(self.llm.kv_len + max_new_tokens)
        tok = out["next_token"]
        new: List[int] = []
        first = True
        if use_graph:
            self.llm.sync_decode_state()
        for i in range(max_new_tokens):
            t = int(tok.item())
            new.append(t)
            if t in stop_ids or i + 1 == max_new_tokens:   # no decode step after the last token
                break
            if use_graph:
                _, tok = self.llm.decode_step_graph(tok if first else None)
                first = False
            else:
                _, _, tok = self.llm.decode_step(tok)
        return new


# ---- synthetic weights at the true architecture (no checkpoint / network on either box) ----------------
def random_weights(cfg: FO1Config, device="cuda", seed: int = 0) -> Dict[str, Dict[str, torch.Tensor]]:
    """Seeded N(0, 0.02^2) linear/conv weights, ~1 norms, generated ON the device in bf16 with the
    checkpoint's key names and shapes (SURVEY §8d 'Synthetic inputs')."""
    g = torch.Generator(device=device).manual_seed(seed)
    bf = torch.bfloat16

    def w(*s, sc=0.02):
        return (torch.randn(*s, generator=g, device=device, dtype=torch.float32) * sc).to(bf)

    def ones(n):
        return (1 + 0.02 * torch.randn(n, generator=g, device=device)).to(bf)

    v, l = cfg.vit, cfg.llm
    d, ff = v.hidden_size, v.intermediate_size
    vit = {"patch_embed.proj.weight": w(d, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)}
    for i in range(v.depth):
        p = f"blocks.{i}."
        vit.update({p + "norm1.weight": ones(d), p + "norm2.weight": ones(d), p + "attn.qkv.weight": w(3 * d, d),
                    p + "attn.qkv.bias": w(3 * d), p + "attn.proj.weight": w(d, d), p + "attn.proj.bias": w(d),
                    p + "mlp.gate_proj.weight": w(ff, d), p + "mlp.gate_proj.bias": w(ff), p + "mlp.up_proj.weight": w(ff, d),
                    p + "mlp.up_proj.bias": w(ff), p + "mlp.down_proj.weight": w(d, ff), p + "mlp.down_proj.bias": w(d)})
    u = v.spatial_merge_size ** 2
    vit.update({"merger.ln_q.weight": ones(d), "merger.mlp.0.weight": w(u * d, u * d), "merger.mlp.0.bias": w(u * d),
                "merger.mlp.2.weight": w(v.out_hidden_size, u * d), "merger.mlp.2.bias": w(v.out_hidden_size)})

    dav = {}
    c = DAVIT_LARGE
    prev = 3
    for i, ch in enumerate(c["dims"]):
        k = c["patch_size"][i]
        dav[f"convs.{i}.proj.weight"], dav[f"convs.{i}.proj.bias"] = w(ch, prev, k, k), w(ch)
        nd = prev if c["patch_prenorm"][i] else ch
        dav[f"convs.{i}.norm.weight"], dav[f"convs.{i}.norm.bias"] = ones(nd), w(nd)
        for j in range(c["depths"][i]):
            for blk, attn in (("spatial_block", "window_attn"), ("channel_block", "channel_attn")):
                p = f"blocks.{i}.{j}.{blk}."
                for cv in ("conv1", "conv2"):
                    dav[p + cv + ".fn.dw.weight"], dav[p + cv + ".fn.dw.bias"] = w(ch, 1, 3, 3, sc=0.05), w(ch)
                dav[p + attn + ".norm.weight"], dav[p + attn + ".norm.bias"] = ones(ch), w(ch)
                dav[p + attn + ".fn.qkv.weight"], dav[p + attn + ".fn.qkv.bias"] = w(3 * ch, ch), w(3 * ch)
                dav[p + attn + ".fn.proj.weight"], dav[p + attn + ".fn.proj.bias"] = w(ch, ch), w(ch)
                dav[p + "ffn.norm.weight"], dav[p + "ffn.norm.bias"] = ones(ch), w(ch)
                dav[p + "ffn.fn.net.fc1.weight"], dav[p + "ffn.fn.net.fc1.bias"] = w(4 * ch, ch), w(4 * ch)
                dav[p + "ffn.fn.net.fc2.weight"], dav[p + "ffn.fn.net.fc2.bias"] = w(ch, 4 * ch), w(ch)
        prev = ch

    fpn = {}
    dim, out = d, 512
    fpn["simfp_1.0.weight"], fpn["simfp_1.0.bias"] = w(dim, dim // 2, 2, 2), w(dim // 2)
    fpn["simfp_1.1.weight"], fpn["simfp_1.1.bias"] = ones(dim // 2), w(dim // 2)
    fpn["simfp_1.3.weight"], fpn["simfp_1.3.bias"] = w(dim // 2, dim // 4, 2, 2), w(dim // 4)
    fpn["simfp_2.0.weight"], fpn["simfp_2.0.bias"] = w(dim, dim // 2, 2, 2), w(dim // 2)
    for name, a, b, cin in (("simfp_1", "4.", "5.", dim // 4), ("simfp_2", "1.", "2.", dim // 2), ("simfp_3", "0.", "1.", dim),
                            ("simfp_4", "1.", "2.", dim)):
        fpn[f"{name}.{a}weight"] = w(out, cin, 1, 1)
        fpn[f"{name}.{a}norm.weight"], fpn[f"{name}.{a}norm.bias"] = ones(out), w(out)
        fpn[f"{name}.{b}weight"] = w(out, out, 3, 3)
        fpn[f"{name}.{b}norm.weight"], fpn[f"{name}.{b}norm.bias"] = ones(out), w(out)

    llm = {"embed_tokens.weight": w(l.vocab_size, l.hidden_size), "norm.weight": ones(l.hidden_size)}
    hq, hk = l.num_heads * l.head_dim, l.num_kv_heads * l.head_dim
    for i in range(l.num_layers):
        p = f"layers.{i}."
        llm.update({p + "input_layernorm.weight": ones(l.hidden_size), p + "post_attention_layernorm.weight": ones(l.hidden_size),
                    p + "self_attn.q_proj.weight": w(hq, l.hidden_size), p + "self_attn.q_proj.bias": w(hq),
                    p + "self_attn.k_proj.weight": w(hk, l.hidden_size), p + "self_attn.k_proj.bias": w(hk),
                    p + "self_attn.v_proj.weight": w(hk, l.hidden_size), p + "self_attn.v_proj.bias": w(hk),
                    p + "self_attn.o_proj.weight": w(l.hidden_size, hq),
                    p + "mlp.gate_proj.weight": w(l.intermediate_size, l.hidden_size),
                    p + "mlp.up_proj.weight": w(l.intermediate_size, l.hidden_size),
                    p + "mlp.down_proj.weight": w(l.hidden_size, l.intermediate_size)})

    proj = {}

    def mlp(prefix, kind, din, dout):
        if kind == "identity":
            return
        if kind == "linear":
            proj[prefix + "weight"], proj[prefix + "bias"] = w(dout, din), w(dout)
            return
        n = int(re.match(r"^mlp(\d+)x_gelu$", kind).group(1))
        for i in range(n):
            proj[f"{prefix}{2 * i}.weight"], proj[f"{prefix}{2 * i}.bias"] = w(dout, din if i == 0 else dout), w(dout)

    mlp("mm_projector.", cfg.mm_projector_type, v.out_hidden_size, l.hidden_size)
    mlp("mm_projector_aux.", cfg.mm_projector_aux_type, cfg.mm_region_hidden_size, l.hidden_size)
    return dict(vit=vit, davit=dav, fpn=fpn, llm=llm, proj=proj)


def synthetic_prompt(n_boxes: int, n_text: int = 60, vocab: int = 151936, seed: int = 0, lead_seed: Optional[int] = None) -> List[int]:
    """Sentinel id sequence of the reference's prompt layout (mm_utils.py:504-521): system/user preamble,
    <image>, then per region one index token + one <regionfeat>, then the question.  No tokenizer is
    available offline, so text ids are seeded randoms (SURVEY §8d 'Prompt').  `lead_seed`: the seed of the 18 preamble ids — several
    prompts over one image carry the SAME preamble (system text, `<|im_start|>user`, `<|vision_start|>`: mm_utils.py:559-575) and
    differ in the regions and the question that follow."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, vocab - 1000, (n_text,), generator=g).tolist()
    pre, post = ids[:18], ids[18:]
    if lead_seed is not None and lead_seed != seed:
        pre = torch.randint(1000, vocab - 1000, (n_text,), generator=torch.Generator().manual_seed(lead_seed)).tolist()[:18]
    seq = pre + [IMAGE_TOKEN_INDEX] + [post[0]]
    for i in range(n_boxes):
        seq += [2000 + i, DEFAULT_REGION_INDEX]
    seq += post[1:]
    return seq
