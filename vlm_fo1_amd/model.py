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
        self.fpn = SimpleFPN(weights["fpn"], device) if cfg.mm_use_simpleFPN_for_vt else None
        self.llm = QwenLLM(cfg.llm, weights["llm"], device, lm_head=weights["llm"].get("lm_head.weight"))
        self.mm_projector = Projector(cfg.mm_projector_type, weights["proj"], "mm_projector.", device)
        self.mm_projector_aux = Projector(cfg.mm_projector_aux_type, weights["proj"], "mm_projector_aux.", device)
        self.hfre = HFREModule(roi_output_size=cfg.mm_roi_output_size, region_feature_dim=cfg.mm_region_hidden_size,
                               apply_position_embedding=cfg.mm_apply_position_embedding, pos_embedding_strategy="bbox_based",
                               use_vision_tower_region_feature=True, region_feature_combination="concat",
                               vision_tower_region_feature_dim=2048 if cfg.mm_use_simpleFPN_for_vt else 4 * cfg.vit.hidden_size,
                               vision_tower_spatial_scale=1 / cfg.vit.patch_size,
                               use_simpleFPN_for_vt=cfg.mm_use_simpleFPN_for_vt, aux_vision_tower_spatial_scale=0.25)
        self._dummy_box = torch.tensor([[0., 10., 0., 10.]], device=self.dev)  # omchat_qwen2_5_vl.py:90-91
        self._graphs = {}
        # Two-stream tower overlap (DaViT || ViT+FPN) is OFF: measured on MI355X / ROCm 7.2 a forked hipGraph replays at
        # 39.7 ms vs 21.9 ms single-stream (cross-stream joins serialise the node launches), see profiles/README.md.
        self._ws_owner = self.llm._ws_owner = object()   # scratch buffers are keyed by this token (ops.workspace_scope)
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
        r._ws_owner = r.llm._ws_owner = object()   # scratch buffers are keyed by this token (ops.workspace_scope)
        r._graphs = {}
        r.stage_hook = None
        return r

    def _mark(self, stage: str):
        """Stage boundary for measurement (bench.py sets `stage_hook` in its eager profiling pass only)."""
        if self.stage_hook is not None:
            self.stage_hook(stage)

    def encode_images(self, pixel_values: torch.Tensor, gh: int, gw: int):
        """-> (image tokens [S/4, d_llm], captured ViT maps (token-major raster))  (encode_images :44-72)."""
        tokens, feats = self.vit.forward(pixel_values, gh, gw, capture="last" if self.fpn is not None else "all")
        self._mark("qwen_vit+merger")
        out = self.mm_projector(tokens)
        self._mark("mm_projector")
        return out, feats

    def encode_regions(self, aux_image: torch.Tensor, boxes: Optional[torch.Tensor], vt_feats: List[torch.Tensor], gh: int, gw: int):
        """-> region tokens [N, d_llm]  (encode_regions :75-108).  boxes: fp32 [N,4] xyxy in aux-image pixels."""
        aux_maps, aux_sizes = self.davit.forward(aux_image)
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
        if self.fpn is not None:
            fpn_maps, fpn_sizes = self.fpn.forward(vt_feats[-1], gh, gw)
            self._mark("simple_fpn")
            fpn_views = [nchw(t, s) for t, s in zip(fpn_maps, fpn_sizes)]
            self.hfre.simple_fpn = lambda x: fpn_views
            vt_in = nchw(vt_feats[-1], (gh, gw))
        else:
            vt_in = [nchw(t, (gh, gw)) for t in vt_feats]
        feat = self.hfre(aux_views, [boxes], vt_in, None, vt_scale=(sw, sh)).squeeze(0)   # fp32 [N, C_region]
        self._mark("hfre_region_pool")
        out = self.mm_projector_aux(feat.to(torch.bfloat16))                               # :106-107
        self._mark("mm_projector_aux")
        return out

    # ---- one image: everything up to the first generated token -----------------------------------
    def _device_prefill(self, pix, gh, gw, aux, boxes, plan_dev, cos, sin, want_regions: bool):
        with ops.workspace_scope(self._ws_owner):
            return self._device_prefill_impl(pix, gh, gw, aux, boxes, plan_dev, cos, sin, want_regions)

    def _device_prefill_impl(self, pix, gh, gw, aux, boxes, plan_dev, cos, sin, want_regions: bool):
        # The two towers are independent until the HFRE, but running DaViT on a side stream INSIDE one pass was measured
        # slower (forked hipGraph replay 39.7 ms vs 21.9 ms on ROCm 7.2); overlap comes from independent images on
        # different streams instead (FO1Engine.replica).
        image_tokens, vt_feats = self.encode_images(pix, gh, gw)
        region_tokens = self.encode_regions(aux, boxes, vt_feats, gh, gw) if want_regions else None
        emb = self.llm.embed_rows(plan_dev, image_tokens, region_tokens)
        self._mark("splice")
        last, logits, tok = self.llm.prefill(emb, None, 0, tables=(cos, sin))
        self._mark("llm_prefill+lm_head+argmax")
        return dict(image_tokens=image_tokens, region_tokens=region_tokens, embeds=emb, last_hidden=last, logits=logits,
                    next_token=tok)

    def prefill(self, input_ids: Sequence[int], pixel_values: torch.Tensor, grid_hw: Tuple[int, int], aux_image: torch.Tensor,
                boxes: Optional[torch.Tensor], use_graph: bool = False):
        """Everything up to the first greedy token.  use_graph=True replays a captured hipGraph of the ~1000
        kernel launches for this (grid, aux size, #boxes, sequence length) signature: index plans / rope tables
        are computed on the host and copied into the graph's static input buffers before the replay."""
        from .llm import mrope_tables
        gh, gw = grid_hw
        m = self.cfg.vit.spatial_merge_size
        n_img = (gh // m) * (gw // m)
        want_regions = (boxes is not None) or any(t == DEFAULT_REGION_INDEX for t in input_ids)
        if boxes is None or boxes.shape[0] == 0:
            boxes = self._dummy_box
        n_reg = boxes.shape[0] if want_regions else 0
        plan, pos, delta = self.llm.plan_inputs(input_ids, n_img, n_reg, (gh // m, gw // m))
        c = self.cfg.llm
        cos, sin = mrope_tables(pos, c.head_dim, c.rope_theta, c.mrope_section)
        boxes = boxes.to(device=self.dev, dtype=torch.float32)
        if not use_graph:
            out = self._device_prefill(pixel_values, gh, gw, aux_image, boxes, plan.to(self.dev), cos.to(self.dev), sin.to(self.dev),
                                       want_regions)
        else:
            key = (gh, gw, tuple(aux_image.shape), n_reg, plan.shape[0], want_regions, pixel_values.dtype, aux_image.dtype)
            ent = self._graphs.get(key)
            if ent is None:
                with ops.graph_lock.capture():   # exclusive: no other thread captures or launches meanwhile
                    # static input buffers must be ordinary tensors even when the caller runs under
                    # torch.inference_mode() (the reference's inference.py:46 does): they are updated in place later
                    with torch.inference_mode(False):
                        st = dict(pix=pixel_values.clone(), aux=aux_image.clone(), boxes=boxes.clone(), plan=plan.clone().to(self.dev),
                                  cos=cos.clone().to(self.dev), sin=sin.clone().to(self.dev))
                    # warm-up on a side stream (allocates every lazily-created scratch buffer), then capture
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        self._device_prefill(st["pix"], gh, gw, st["aux"], st["boxes"], st["plan"], st["cos"], st["sin"], want_regions)
                    torch.cuda.current_stream().wait_stream(s)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL watchdog threads may touch the runtime meanwhile
                        res = self._device_prefill(st["pix"], gh, gw, st["aux"], st["boxes"], st["plan"], st["cos"], st["sin"], want_regions)
                    ent = (g, st, res, [])
                    self._graphs[key] = ent
            g, st, res, keep = ent
            st["pix"].copy_(pixel_values, non_blocking=pixel_values.is_cuda)   # device -> device; host tensors upload blocking
            st["aux"].copy_(aux_image, non_blocking=aux_image.is_cuda)
            st["boxes"].copy_(boxes, non_blocking=boxes.is_cuda)
            # Host-built inputs: async uploads from pageable tensors.  The runtime stages the bytes at call time (measured: an
            # event-guarded pinned buffer or a blocking copy serialises graph launch and execution, 31 vs 19 ms per image);
            # the last few source tensors are kept alive anyway so their memory cannot be re-used under a pending copy.
            for k, src in (("plan", plan), ("cos", cos), ("sin", sin)):
                st[k].copy_(src, non_blocking=True)
            keep.append((plan, cos, sin))
            if len(keep) > 8:
                del keep[0]
            with ops.graph_lock.replay():
                g.replay()
            out = dict(res)
        self.llm.kv_len = plan.shape[0]
        self.llm.rope_delta = delta
        out["position_ids"] = pos
        out["rope_delta"] = delta
        return out

    def generate(self, input_ids: Sequence[int], pixel_values, grid_hw, aux_image, boxes, max_new_tokens: int = 512,
                 stop_ids: Sequence[int] = (), use_graph: bool = False) -> List[int]:
        """Greedy decode (do_sample=False in every reference caller: mm_utils.py:640-654)."""
        out = self.prefill(input_ids, pixel_values, grid_hw, aux_image, boxes, use_graph=use_graph)
        tok = out["next_token"]
        new: List[int] = []
        first = True
        if use_graph:
            self.llm.sync_decode_state()
        for _ in range(max_new_tokens):
            t = int(tok.item())
            new.append(t)
            if t in stop_ids:
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


def synthetic_prompt(n_boxes: int, n_text: int = 60, vocab: int = 151936, seed: int = 0) -> List[int]:
    """Sentinel id sequence of the reference's prompt layout (mm_utils.py:504-521): system/user preamble,
    <image>, then per region one index token + one <regionfeat>, then the question.  No tokenizer is
    available offline, so text ids are seeded randoms (SURVEY §8d 'Prompt')."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, vocab - 1000, (n_text,), generator=g).tolist()
    pre, post = ids[:18], ids[18:]
    seq = pre + [IMAGE_TOKEN_INDEX] + [post[0]]
    for i in range(n_boxes):
        seq += [2000 + i, DEFAULT_REGION_INDEX]
    seq += post[1:]
    return seq
