"""Stage-level C-ABI driven from Python: builds the fo1_*_weights_t / plan / cache structs of include/fo1.h from the engine's
modules and calls fo1_vit_forward / fo1_llm_prefill / fo1_llm_decode_step.  This is what a non-Python host does with the same
pointers (INTEGRATION.md shows the C side); here it exists so that tests can hold the stage entries bit-identical to the Python
orchestration (tests/test_stage_abi_gpu.py), and as an alternative dispatch for the engine (FO1_STAGE_ABI=1)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import lib as _lib
from . import ops


import os

ENABLED = os.environ.get("FO1_STAGE_ABI") == "1"   # route QwenViT._forward / QwenLLM.prefill_packed / BatchDecoder steps through the C stage entries


def enabled() -> bool:
    return ENABLED


def vit_stage(vit) -> "VitStage":
    st = getattr(vit, "_stage", None)
    if st is None:
        st = vit._stage = VitStage(vit)
    return st


def llm_stage(llm) -> "LlmStage":
    st = getattr(llm, "_stage", None)
    if st is None or st.llm is not llm or st.W.embed != llm.embed.data_ptr():     # (a replica() copies the attribute)
        st = llm._stage = LlmStage(llm)
    return st


def davit_stage(davit) -> "DavitStage":
    st = getattr(davit, "_stage", None)
    if st is None:
        st = davit._stage = DavitStage(davit)
    return st


def fpn_stage(fpn) -> "FpnStage":
    st = getattr(fpn, "_stage", None)
    if st is None:
        st = fpn._stage = FpnStage(fpn)
    return st


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class VitStage:
    def __init__(self, vit):
        self.vit = vit
        c = vit.cfg
        self._blocks = (_lib.VitBlock * c.depth)()
        for i, w in enumerate(vit.blocks):
            for k in ("n1", "n2", "wqkv", "bqkv", "wo", "bo", "wgu", "bgu", "wd", "bd"):
                setattr(self._blocks[i], k, w[k].data_ptr())
            for k in ("wqkv_hm", "bqkv_hm"):      # head-major copy for the fused q/k/v epilogue (ABI 7), when the tower built one
                setattr(self._blocks[i], k, w[k].data_ptr() if w.get(k) is not None else None)
        W = _lib.VitWeights()
        W.depth, W.hidden, W.n_heads, W.ff_padded = c.depth, c.hidden_size, c.num_heads, vit.ffp
        W.k_in, W.k_in_padded, W.merge, W.out_hidden = vit.k_in, vit.k_in_p, c.spatial_merge_size, c.out_hidden_size
        W.n_fullatt = len(c.fullatt_block_indexes)
        for i, b in enumerate(c.fullatt_block_indexes):
            W.fullatt[i] = b
        W.patch_w = vit.patch_w.data_ptr()
        W.blocks = ctypes.cast(self._blocks, ctypes.POINTER(_lib.VitBlock))
        W.ln_q, W.m0w, W.m0b, W.m2w, W.m2b = (t.data_ptr() for t in (vit.ln_q, vit.m0w, vit.m0b, vit.m2w, vit.m2b))
        self.W = W

    def plan_struct(self, g) -> "_lib.VitPlan":
        d = self.vit.cfg.hidden_size
        P = _lib.VitPlan()
        P.S = g.S
        P.plan_in, P.plan_raster, P.plan_tokens = g.plan_in.data_ptr(), g.plan_raster.data_ptr(), g.plan_tokens.data_ptr()
        P.cos, P.sin = g.cos.data_ptr(), g.sin.data_ptr()
        P.items_win, P.n_items_win = g.items_win.data_ptr(), g.items_win.shape[0]
        P.q_block_win = 0 if getattr(g.items_win, "single_tile", False) else getattr(g.items_win, "q_block", 64)      # 0: fo1_attention_windows_bf16
        P.items_full, P.n_items_full, P.q_block_full = g.items_full.data_ptr(), g.items_full.shape[0], getattr(g.items_full, "q_block", 64)
        if getattr(g, "cu_window", None) is not None:
            win, full = list(zip(g.cu_window[:-1], g.cu_window[1:])), [(0, g.S)]
        else:
            win, full = g.win_segments, g.full_segments
        P.flops_win = 4.0 * d * sum((b - a) ** 2 for a, b in win)
        P.flops_full = 4.0 * d * sum((b - a) ** 2 for a, b in full)
        return P

    def forward(self, pixel_values: torch.Tensor, g, capture: str = "all"):
        """Same contract as QwenViT._forward (pixel rows bf16 [S, 1176], plan) -> (tokens, feature maps)."""
        L = _lib.load()
        c = self.vit.cfg
        dev = pixel_values.device
        pix = pixel_values.to(torch.bfloat16).contiguous()
        u = c.spatial_merge_size ** 2
        tokens = torch.empty(g.S // u, c.out_hidden_size, dtype=torch.bfloat16, device=dev)
        nf = len(c.fullatt_block_indexes)
        want = list(range(nf)) if capture == "all" else ([] if capture == "none" else [nf - 1])    # null slots are not written
        feats = [torch.empty(g.S, c.hidden_size, dtype=torch.bfloat16, device=dev) for _ in want]
        arr = (ctypes.c_void_p * nf)()
        for t, k in zip(feats, want):
            arr[k] = t.data_ptr()
        need = L.fo1_vit_workspace_bytes(ctypes.byref(self.W), g.S)
        ws = ops._workspace("stage_vit", dev, need)
        P = self.plan_struct(g)
        rc = L.fo1_vit_forward(ctypes.byref(self.W), ctypes.byref(P), pix.data_ptr(), pix.stride(0), tokens.data_ptr(), arr, ws.data_ptr(), ws.numel(),
                               _lib.current_stream_ptr())
        _lib.check(rc, "fo1_vit_forward")
        self._keep = (pix, P)
        return tokens, feats


class LlmStage:
    def __init__(self, llm):
        self.llm = llm
        c = llm.cfg
        self._layers = (_lib.LlmLayer * c.num_layers)()
        for i, w in enumerate(llm.layers):
            for k in ("ln1", "ln2", "wqkv", "bqkv", "wo", "wgu", "wdown"):
                setattr(self._layers[i], k, w[k].data_ptr())
        W = _lib.LlmWeights()
        W.n_layers, W.hidden, W.n_heads, W.n_kv_heads, W.head_dim = c.num_layers, c.hidden_size, c.num_heads, c.num_kv_heads, c.head_dim
        W.intermediate, W.vocab, W.rms_eps = llm.layers[0]["wdown"].shape[1], llm.lm_head.shape[0], c.rms_norm_eps
        W.layers = ctypes.cast(self._layers, ctypes.POINTER(_lib.LlmLayer))
        W.embed, W.final_norm, W.lm_head = llm.embed.data_ptr(), llm.norm.data_ptr(), llm.lm_head.data_ptr()
        self.W = W

    @staticmethod
    def cache_struct(k: torch.Tensor, vt: torch.Tensor) -> "_lib.KvCache":
        """k [layers, n_kv, rows, hd], vt [layers, n_kv*hd, rows]"""
        C = _lib.KvCache()
        C.k, C.k_layer_stride, C.k_head_stride = k.data_ptr(), k.stride(0), k.stride(1)
        C.vt, C.vt_layer_stride, C.vt_row_stride = vt.data_ptr(), vt.stride(0), vt.stride(1)
        C.capacity = k.shape[2]
        return C

    def prefill_packed(self, embeds: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, seqs, last_plan: torch.Tensor, want_hidden: bool = False):
        """Same contract as QwenLLM.prefill_packed -> (last hidden [B,d], logits [B,V], next ids [B]) (+ final hidden rows)."""
        L = _lib.load()
        llm, c = self.llm, self.llm.cfg
        dev = embeds.device
        items, flops, prefix_ranges = llm.packed_items(seqs)
        assert prefix_ranges is None, "fo1_llm_prefill takes plain work items (QwenLLM.prefill_packed keeps shared-prefix passes on the primitive path)"
        R, B = embeds.shape[0], last_plan.shape[0]
        last = torch.empty(B, c.hidden_size, dtype=torch.bfloat16, device=dev)
        logits = torch.empty(B, self.W.vocab, dtype=torch.bfloat16, device=dev)
        toks = torch.empty(B, dtype=torch.int32, device=dev)
        hidden = torch.empty(R, c.hidden_size, dtype=torch.bfloat16, device=dev) if want_hidden else None
        need = L.fo1_llm_prefill_workspace_bytes(ctypes.byref(self.W), R, B)
        ws = ops._workspace("stage_llm", dev, need)
        C = self.cache_struct(llm.kcache, llm.vtcache)
        rc = L.fo1_llm_prefill(ctypes.byref(self.W), ctypes.byref(C), embeds.data_ptr(), embeds.stride(0), cos.data_ptr(), sin.data_ptr(), R, 0,
                               items.data_ptr(), items.shape[0], getattr(items, "q_block", 64), float(flops), last_plan.data_ptr(), B, _p(hidden),
                               last.data_ptr(), logits.data_ptr(), toks.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "fo1_llm_prefill")
        return (last, logits, toks, hidden) if want_hidden else (last, logits, toks)

    def decode_step(self, dec) -> torch.Tensor:
        """One step of a started BatchDecoder `dec` through fo1_llm_decode_step; returns the logits [B, V]."""
        L = _lib.load()
        llm, B = self.llm, dec.B
        dev = dec.state.device
        logits = torch.empty(B, self.W.vocab, dtype=torch.bfloat16, device=dev)
        need = L.fo1_llm_decode_workspace_bytes(ctypes.byref(self.W), B, dec.slot)
        ws = ops._workspace("stage_decode", dev, need)
        C = self.cache_struct(dec.dk, dec.dvt)
        rc = L.fo1_llm_decode_step(ctypes.byref(self.W), ctypes.byref(C), llm.rope_cos.data_ptr(), llm.rope_sin.data_ptr(), dec.state.data_ptr(),
                                   dec.plan.data_ptr(), dec.ids.data_ptr(), dec.ids.shape[1], dec.stop.data_ptr() if dec.n_stop else None, dec.n_stop,
                                   dec.done.data_ptr(), B, dec.slot, dec.kv_bucket(), logits.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "fo1_llm_decode_step")
        return logits


class DavitStage:
    def __init__(self, davit):
        self.davit = davit
        cfg = davit.cfg
        W = _lib.DavitWeights()
        W.n_stages, W.window = len(cfg["dims"]), cfg["window"]
        self._blocks = []
        for i, C in enumerate(cfg["dims"]):
            cv = davit.convs[i]
            S = W.stages[i]
            S.dim, S.heads, S.depth = C, cfg["heads"][i], cfg["depths"][i]
            S.kernel, S.stride, S.pad = cfg["patch_size"][i], cfg["patch_stride"][i], cfg["patch_padding"][i]
            S.prenorm, S.K_padded = int(cfg["patch_prenorm"][i]), cv["Kp"]
            S.conv_w, S.conv_b, S.norm_w, S.norm_b = cv["w"].data_ptr(), cv["b"].data_ptr(), cv["nw"].data_ptr(), cv["nb"].data_ptr()
            arr = (_lib.DavitBlock * S.depth)()
            for j, blk in enumerate(davit.blocks[i]):
                for half, name in ((arr[j].spatial, "spatial_block"), (arr[j].channel, "channel_block")):
                    for k in _lib._HALF:
                        setattr(half, k, blk[name][k].data_ptr())
            self._blocks.append(arr)
            S.blocks = ctypes.cast(arr, ctypes.POINTER(_lib.DavitBlock))
        self.W = W

    def forward(self, img: torch.Tensor):
        """Same contract as DaViT.forward: img [B,3,H,W] -> ([4 maps], [(H_i, W_i)])."""
        L = _lib.load()
        d, cfg = self.davit, self.davit.cfg
        if img.dim() == 3:
            img = img.unsqueeze(0)
        img = img.contiguous()
        B, _, H, W = img.shape
        ws = cfg["window"]
        P = _lib.DavitPlan()
        P.H, P.W, P.batch = H, W, B
        sizes, outs, keep = [], [], []
        h, w = H, W
        for i, C in enumerate(cfg["dims"]):
            k, s, p = cfg["patch_size"][i], cfg["patch_stride"][i], cfg["patch_padding"][i]
            h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            sizes.append((h, w))
            n_win = B * ((h + ws - 1) // ws) * ((w + ws - 1) // ws)
            items = d._window_items(n_win, ws * ws, cfg["heads"][i])
            keep.append(items)
            P.items[i], P.n_items[i], P.q_block[i] = items.data_ptr(), items.shape[0], getattr(items, "q_block", 64)
            outs.append(torch.empty(B * h * w, C, dtype=torch.bfloat16, device=img.device))
        arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in outs])
        need = L.fo1_davit_workspace_bytes(ctypes.byref(self.W), ctypes.byref(P))
        wsb = ops._workspace("stage_davit", img.device, need)
        rc = L.fo1_davit_forward(ctypes.byref(self.W), ctypes.byref(P), img.data_ptr(), 1 if img.dtype == torch.float32 else 0, arr, wsb.data_ptr(), wsb.numel(),
                                 _lib.current_stream_ptr())
        _lib.check(rc, "fo1_davit_forward")
        self._keep = (img, keep)
        return outs, sizes


class FpnStage:
    def __init__(self, fpn):
        self.fpn = fpn
        W = _lib.FpnWeights()
        W.c_in = fpn.t1a[0].shape[1]
        W.c_up1, W.c_up2, W.c_out = fpn.t1a[2], fpn.t1b[2], fpn.heads[0]["w1"].shape[0]
        assert fpn.t2[2] == W.c_up1
        W.t1a_w, W.t1a_b = fpn.t1a[0].data_ptr(), fpn.t1a[1].data_ptr()
        W.t1_ln_w, W.t1_ln_b = fpn.t1_ln[0].data_ptr(), fpn.t1_ln[1].data_ptr()
        W.t1b_w, W.t1b_b = fpn.t1b[0].data_ptr(), fpn.t1b[1].data_ptr()
        W.t2_w, W.t2_b = fpn.t2[0].data_ptr(), fpn.t2[1].data_ptr()
        for i, h in enumerate(fpn.heads):
            W.heads[i].w1, W.heads[i].n1_w, W.heads[i].n1_b = h["w1"].data_ptr(), h["n1"][0].data_ptr(), h["n1"][1].data_ptr()
            W.heads[i].w3, W.heads[i].n3_w, W.heads[i].n3_b = h["w3"].data_ptr(), h["n3"][0].data_ptr(), h["n3"][1].data_ptr()
        self.W = W

    def forward(self, x: torch.Tensor, H: int, W: int, batch: int = 1):
        L = _lib.load()
        co = self.W.c_out
        sizes = [(4 * H, 4 * W), (2 * H, 2 * W), (H, W), (H // 2, W // 2)]
        outs = [torch.empty(batch * h * w, co, dtype=torch.bfloat16, device=x.device) for h, w in sizes]
        arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in outs])
        x = x.contiguous()
        need = L.fo1_simplefpn_workspace_bytes(ctypes.byref(self.W), H, W, batch)
        wsb = ops._workspace("stage_fpn", x.device, need)
        rc = L.fo1_simplefpn_forward(ctypes.byref(self.W), x.data_ptr(), H, W, batch, arr, wsb.data_ptr(), wsb.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "fo1_simplefpn_forward")
        self._keep = x
        return outs, sizes


def projector_forward(layers, x: torch.Tensor) -> torch.Tensor:
    """layers: [(weight [out, in], bias [out])] of an mlpN_gelu connector."""
    L = _lib.load()
    P = _lib.ProjectorW()
    P.n_layers = len(layers)
    P.dims[0] = layers[0][0].shape[1]
    for i, (w, b) in enumerate(layers):
        P.dims[i + 1] = w.shape[0]
        P.w[i], P.b[i] = w.data_ptr(), b.data_ptr()
    x = x.contiguous()
    out = torch.empty(x.shape[0], P.dims[P.n_layers], dtype=torch.bfloat16, device=x.device)
    need = L.fo1_projector_workspace_bytes(ctypes.byref(P), x.shape[0])
    wsb = ops._workspace("stage_proj", x.device, need)
    rc = L.fo1_projector_forward(ctypes.byref(P), x.data_ptr(), x.stride(0), x.shape[0], out.data_ptr(), out.stride(0), wsb.data_ptr(), wsb.numel(),
                                 _lib.current_stream_ptr())
    _lib.check(rc, "fo1_projector_forward")
    return out
