"""Qwen2.5-VL language-model half of the hot path on the MI355X engine.

Host code only orchestrates: every tensor op is a libfo1hip.so kernel (vlm_fo1_amd/ops.py).  Index
bookkeeping the reference does with device syncs (`get_rope_index`, sentinel search,
modeling_qwen2_5_vl.py:1546-1701, omchat_qwen2_5_vl.py:318-319) is done on the host from the
sentinel positions and the image grid before any launch.

Weights: fused at load time — q/k/v -> one [2560, 2048] GEMM, gate/up -> one [22016, 2048] GEMM
(rows interleaved [gate 16 | up 16 | ...] so SwiGLU runs in the GEMM epilogue) — checkpoint key names as in the reference state dict.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

IMAGE_TOKEN_INDEX = -200
DEFAULT_REGION_INDEX = -300


@dataclass
class LLMConfig:
    hidden_size: int = 2048
    num_layers: int = 36
    num_heads: int = 16
    num_kv_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 11008
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    max_seq: int = 8192


_ROPE_INDEX_CACHE: Dict[tuple, tuple] = {}


def rope_index_host(n_before: int, grid_hw_merged: Tuple[int, int], n_after: int):
    """[3, L] int64 position ids for <text><image gh x gw><text> and the rope delta
    (reference get_rope_index :1630-1697, single image, t = 1).  Memoised: a pure function of the prompt structure."""
    key = (int(n_before), int(grid_hw_merged[0]), int(grid_hw_merged[1]), int(n_after))
    hit = _ROPE_INDEX_CACHE.get(key)
    if hit is None:
        if len(_ROPE_INDEX_CACHE) >= 1024:
            _ROPE_INDEX_CACHE.pop(next(iter(_ROPE_INDEX_CACHE)))
        hit = _ROPE_INDEX_CACHE[key] = _rope_index_host(n_before, grid_hw_merged, n_after)
    return hit


def _rope_index_host(n_before: int, grid_hw_merged: Tuple[int, int], n_after: int):
    gh, gw = grid_hw_merged
    pre = torch.arange(n_before).view(1, -1).expand(3, -1)
    t_idx = torch.zeros(gh * gw, dtype=torch.long)
    h_idx = torch.arange(gh).view(-1, 1).expand(-1, gw).flatten()
    w_idx = torch.arange(gw).view(1, -1).expand(gh, -1).flatten()
    img = torch.stack([t_idx, h_idx, w_idx]) + n_before
    nxt = int(img.max()) + 1 if gh * gw > 0 else n_before
    post = torch.arange(n_after).view(1, -1).expand(3, -1) + nxt
    pos = torch.cat([pre, img, post], dim=1)
    return pos, int(pos.max()) + 1 - pos.shape[1]


_MROPE_CACHE: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def mrope_tables(pos: torch.Tensor, head_dim: int, theta: float, sections: Sequence[int]):
    """pos [3, L] (host) -> cos, sin [L, head_dim] bf16 (host): fp32 tables, section select, cast
    (reference :609-624, :675-681).  Memoised on the position ids: they depend only on the prompt structure (tokens before the
    image, merged grid, tokens after), which repeats from request to request, and the dozen small torch CPU ops below cost
    ~10 ms per call on a many-core host (intra-op thread-pool wake-ups) — more than half a GPU pass."""
    key = (pos.shape[1], head_dim, float(theta), tuple(sections), pos.numpy().tobytes())
    hit = _MROPE_CACHE.get(key)
    if hit is not None:
        return hit
    out = _mrope_tables(pos, head_dim, theta, sections)
    if len(_MROPE_CACHE) >= 256:
        _MROPE_CACHE.pop(next(iter(_MROPE_CACHE)))
    _MROPE_CACHE[key] = out
    return out


def _mrope_tables(pos: torch.Tensor, head_dim: int, theta: float, sections: Sequence[int]):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = pos.float()[:, :, None] * inv[None, None, :]
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos(), emb.sin()
    sec = list(sections) * 2
    cs = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sn = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cs.to(torch.bfloat16).contiguous(), sn.to(torch.bfloat16).contiguous()


class QwenLLM:
    def __init__(self, cfg: LLMConfig, state: Dict[str, torch.Tensor], device, lm_head: Optional[torch.Tensor] = None):
        self.cfg = cfg
        self.dev = torch.device(device)
        bf = torch.bfloat16

        def dv(t):
            return t.to(device=self.dev, dtype=bf).contiguous()

        self.embed = dv(state["embed_tokens.weight"])
        self.lm_head = dv(lm_head) if lm_head is not None else self.embed  # tied embeddings
        self.norm = dv(state["norm.weight"])
        self.layers = []
        for i in range(cfg.num_layers):
            p = f"layers.{i}."
            a = p + "self_attn."
            self.layers.append(dict(
                ln1=dv(state[p + "input_layernorm.weight"]),
                ln2=dv(state[p + "post_attention_layernorm.weight"]),
                wqkv=dv(torch.cat([state[a + "q_proj.weight"], state[a + "k_proj.weight"], state[a + "v_proj.weight"]], 0)),
                bqkv=dv(torch.cat([state[a + "q_proj.bias"], state[a + "k_proj.bias"], state[a + "v_proj.bias"]], 0)),
                wo=dv(state[a + "o_proj.weight"]),
                wgu=dv(ops.interleave_gate_up(state[p + "mlp.gate_proj.weight"], state[p + "mlp.up_proj.weight"])),
                wdown=dv(state[p + "mlp.down_proj.weight"]),
            ))
        c = cfg
        # KV cache: K [layer][kv_head][pos][head_dim]; V^T [layer][kv_head*head_dim][pos] (zeroed: the
        # attention kernel may read up to 3 finite columns past kv_end)
        self.capacity = c.max_seq
        self.cache_epoch = 0        # bumped whenever the caches are re-allocated: captured graphs hold the old pointers
        self.kcache = torch.zeros(c.num_layers, c.num_kv_heads, c.max_seq, c.head_dim, dtype=bf, device=self.dev)
        self.vtcache = torch.zeros(c.num_layers, c.num_kv_heads * c.head_dim, c.max_seq, dtype=bf, device=self.dev)
        self.kv_len = 0
        self.rope_delta = 0
        self._item_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        # ---- decode-step graph state (fo1_decode_advance, include/fo1.h) ----
        # text positions have t = h = w, so one [max_seq, head_dim] table indexed by (cache position + rope delta)
        # covers every decode step; it is built on the host with the prefill's exact code path
        p = torch.arange(c.max_seq).view(1, -1).expand(3, -1)
        cos_t, sin_t = mrope_tables(p, c.head_dim, c.rope_theta, c.mrope_section)
        self.rope_cos, self.rope_sin = cos_t.to(self.dev), sin_t.to(self.dev)
        self.dstate = torch.zeros(8, dtype=torch.int32, device=self.dev)       # [pos, rope_row, -, -, item(4)]
        self.dplan = torch.zeros(1, 2, dtype=torch.int32, device=self.dev)     # gather plan of the one new token: (0, token id)
        self._dgraph = None
        self._dstate_keep = None
        self.rope_epoch = 0         # bumped whenever the rope tables are re-allocated (grow_rope): step graphs key on it
        self._rope_retired = []     # replaced tables, never freed (shared by the replicas: see grow_rope)
        self._ws_owner = ops.new_owner(self)   # scratch-buffer key (ops.workspace_scope); FO1Engine overrides it with its own token

    def reserve(self, n_positions: int) -> bool:
        """Make room for n_positions cache rows (prompt + the tokens a request may generate; a packed batch needs the sum).
        Grows geometrically and keeps the cached prefix; returns True when the caches moved (graphs that captured the old
        pointers — this object's decode graph, the engine's prefill graphs — are dropped by the epoch bump).  The reference's
        HF cache grows without bound (max_position_embeddings 32768+), so a long generation must not fail on a fixed size."""
        if n_positions <= self.capacity:
            return False
        c = self.cfg
        cap = self.capacity
        while cap < n_positions:
            cap *= 2
        bf = torch.bfloat16
        k = torch.zeros(c.num_layers, c.num_kv_heads, cap, c.head_dim, dtype=bf, device=self.dev)
        v = torch.zeros(c.num_layers, c.num_kv_heads * c.head_dim, cap, dtype=bf, device=self.dev)
        k[:, :, :self.capacity] = self.kcache
        v[:, :, :self.capacity] = self.vtcache
        self.kcache, self.vtcache, self.capacity = k, v, cap
        # (the text-position rope table is NOT grown here: the prefill does not read it — packed passes bring their own tables — and a
        # worker thread's reserve() must not re-allocate a table that the decode pool's graphs, replaying on another stream, still read
        # (ADVICE r4).  The decode paths grow it when THEY need the rows: _check_room, BatchDecoder._ensure, DecodePool.)
        self._dgraph = None
        self._item_cache = {}
        self.cache_epoch += 1
        torch.cuda.current_stream().synchronize()
        return True

    def grow_rope(self, rows: int) -> None:
        """Text-position rope table (t = h = w) with at least `rows` rows.  Re-allocating it invalidates every captured graph that read
        the old table: the single-sequence decode graph is dropped here, callers drop their own (BatchDecoder._ensure)."""
        c = self.cfg
        if self.rope_cos.shape[0] >= rows:
            return
        p = torch.arange(rows).view(1, -1).expand(3, -1)
        cos_t, sin_t = mrope_tables(p, c.head_dim, c.rope_theta, c.mrope_section)
        # the old tables stay allocated for the life of the process (0.5 KB per row, geometric growth: at most twice the final size):
        # a step graph captured on another engine replica / the decode pool may still be replaying with their pointers on its own
        # stream, and nothing here can wait for it (the epoch / data_ptr graph keys only protect the NEXT capture)
        self._rope_retired.append((self.rope_cos, self.rope_sin))
        self.rope_cos, self.rope_sin = cos_t.to(self.dev), sin_t.to(self.dev)
        self._dgraph = None
        self.rope_epoch += 1

    def replica(self) -> "QwenLLM":
        """Same weights and rope tables (shared, read-only), private per-request state: KV cache, decode state, decode graph.
        For several requests in flight on different streams (FO1Engine.replica)."""
        import copy
        r = copy.copy(self)
        c = self.cfg
        r.kcache = torch.zeros_like(self.kcache)
        r.vtcache = torch.zeros_like(self.vtcache)
        r.kv_len = 0
        r.rope_delta = 0
        r.dstate = torch.zeros_like(self.dstate)
        r.dplan = torch.zeros_like(self.dplan)
        r._dgraph = None
        r._dstate_keep = None
        r._item_cache = {}
        r._ws_owner = ops.new_owner(r)
        # the zero-fills above ran on the creating thread's current stream; the replica is used from another thread / stream
        torch.cuda.current_stream().synchronize()
        return r

    # ---- splice ------------------------------------------------------------------------------
    def plan_inputs(self, input_ids: Sequence[int], n_img: int, n_regions: int, grid_hw_merged: Tuple[int, int]):
        """HOST part of the splice: sentinel ids -> (gather plan int32 [L',2] (cpu), pos [3,L'] (cpu), rope delta).
        One image per prompt (what prepare_inputs produces).  Vectorised: at > 100 images/s per GPU a Python loop over the ~650 rows of
        every prompt (3-4 ms each) is most of a worker thread's time under the GIL (profiles/r04_driver_level_host_profile_first.log)."""
        import numpy as np
        ids = np.asarray(input_ids, dtype=np.int64).reshape(-1)
        img_at = np.flatnonzero(ids == IMAGE_TOKEN_INDEX)
        if img_at.size > 1:
            raise ValueError("more than one <image> sentinel in the prompt")
        if img_at.size == 0:
            raise ValueError("prompt has no <image> sentinel")
        is_reg = ids == DEFAULT_REGION_INDEX
        if int(is_reg.sum()) > n_regions:
            # same failure the reference raises at omchat_qwen2_5_vl.py:361
            raise IndexError("prompt has more <regionfeat> placeholders than region features")
        text = ~is_reg
        text[img_at[0]] = False
        bad = ids[text]
        if bad.size and (int(bad.min()) < 0 or int(bad.max()) >= self.cfg.vocab_size):   # torch's embedding lookup raises the same way in the reference
            t = int(bad[(bad < 0) | (bad >= self.cfg.vocab_size)][0])
            raise IndexError(f"token id {t} is outside the embedding table (vocab_size {self.cfg.vocab_size})")
        if grid_hw_merged[0] * grid_hw_merged[1] != n_img:
            # reference modeling_qwen2_5_vl.py:1797-1800
            raise ValueError(f"Image features and image tokens do not match: tokens: {grid_hw_merged[0] * grid_hw_merged[1]}, features {n_img}")
        n_before = int(img_at[0])
        kind = np.where(is_reg, 2, 0).astype(np.int32)
        val = np.where(is_reg, np.cumsum(is_reg) - 1, ids).astype(np.int32)
        plan = np.empty((ids.size - 1 + n_img, 2), dtype=np.int32)
        plan[:n_before, 0], plan[:n_before, 1] = kind[:n_before], val[:n_before]
        plan[n_before:n_before + n_img, 0] = 1
        plan[n_before:n_before + n_img, 1] = np.arange(n_img, dtype=np.int32)
        plan[n_before + n_img:, 0], plan[n_before + n_img:, 1] = kind[n_before + 1:], val[n_before + 1:]
        n_after = plan.shape[0] - n_before - n_img
        pos, delta = rope_index_host(n_before, grid_hw_merged, n_after)
        return torch.from_numpy(plan), pos, delta

    def decode_weight_tensors(self):
        """The tensors one decode step streams once (projection weights + biases, norms, lm_head): the algorithmic bytes of the step's
        HBM roofline (bench.py `decode.roofline`)."""
        out = [self.lm_head, self.norm]
        for w in self.layers:
            out += [w["wqkv"], w["bqkv"], w["wo"], w["wgu"], w["wdown"], w["ln1"], w["ln2"]]
        return out

    def embed_rows(self, plan_dev: torch.Tensor, image_tokens: torch.Tensor, region_tokens: Optional[torch.Tensor]):
        """DEVICE part of the splice: one row gather over (embed table | image tokens | region tokens)."""
        return ops.gather_rows(plan_dev, self.cfg.hidden_size, self.embed, image_tokens, region_tokens)

    def build_inputs(self, input_ids: Sequence[int], image_tokens: torch.Tensor, region_tokens: Optional[torch.Tensor],
                     grid_hw_merged: Tuple[int, int]):
        """Sentinel ids -> (embeds [L', d] on device, pos [3, L'] host, rope delta)."""
        n_reg = 0 if region_tokens is None else region_tokens.shape[0]
        plan, pos, delta = self.plan_inputs(input_ids, image_tokens.shape[0], n_reg, grid_hw_merged)
        return self.embed_rows(plan.to(self.dev), image_tokens, region_tokens), pos, delta

    # ---- transformer -------------------------------------------------------------------------
    def _forward(self, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos0: int, collect: Optional[list] = None,
                 items: Optional[torch.Tensor] = None, flops: Optional[float] = None, prefix_ranges: Optional[torch.Tensor] = None):
        c = self.cfg
        L = x.shape[0]
        H, KV, HD = c.num_heads, c.num_kv_heads, c.head_dim
        kv_end = pos0 + L
        if kv_end > self.capacity:
            raise ValueError(f"sequence {kv_end} exceeds the KV cache ({self.capacity}); call reserve() first")
        if items is None:
            items = self._items(pos0, kv_end)
        scale = 1.0 / math.sqrt(HD)
        if flops is None:
            flops = 4.0 * H * HD * (L * (pos0 + (L + 1) / 2.0))
        # (the C entry's own preconditions, mirrored — csrc/stages.hip takes the two-launch path on the same terms: N a multiple of the 256-column
        # tile, 16-byte aligned rope tables)
        fused = (HD == 128 and pos0 % 8 == 0 and self.capacity % 8 == 0 and ((H + 2 * KV) * HD) % 256 == 0
                 and cos.data_ptr() % 16 == 0 and sin.data_ptr() % 16 == 0 and cos.is_contiguous() and sin.is_contiguous()
                 and ops.qkv_fused_for(L, (H + 2 * KV) * HD, c.hidden_size) and not ops.fp8_routed(self.layers[0]["wqkv"], L))
        for li, w in enumerate(self.layers):
            if fused:
                # q/k/v projection with mRoPE + K append + V^T in the GEMM's epilogue (ops.qkv_proj_rope mode 0): same bits, one launch less and no
                # second pass over the [L, 2560] activation
                qkv = ops.qkv_proj_rope(ops.rmsnorm(x, w["ln1"], c.rms_norm_eps), w["wqkv"], w["bqkv"], 0, H, KV, cos, sin, self.kcache[li], pos0, self.vtcache[li])
            else:
                qkv = ops.norm_linear(x, w["ln1"], c.rms_norm_eps, w["wqkv"], w["bqkv"])
                ops.qkv_post_llm(qkv, H, KV, HD, cos, sin, self.kcache[li], self.vtcache[li], pos0)   # mRoPE + K append + V^T, one launch
            att = ops.attention_strided(qkv[:, :H * HD], q_row0=pos0, k=self.kcache[li], vt=self.vtcache[li], items=items,
                                        n_q_heads=H, n_kv_heads=KV, head_dim=HD, scale=scale, causal=True, flops=flops, prefix_ranges=prefix_ranges)
            x = ops.gemm(att, w["wo"], residual=x)
            a = ops.norm_linear(x, w["ln2"], c.rms_norm_eps, w["wgu"], act=ops.ACT_SWIGLU16)   # gate/up GEMM, SwiGLU in its epilogue
            x = ops.gemm(a, w["wdown"], residual=x)
            if collect is not None:
                collect.append(x)
        return x

    def _items(self, pos0: int, kv_end: int) -> torch.Tensor:
        key = (pos0, kv_end)
        it = self._item_cache.get(key)
        if it is None:
            # never evict: a captured prefill graph may hold this tensor's pointer (entries are a few hundred bytes each)
            blk = ops.pick_q_block([(pos0, kv_end)], self.cfg.num_heads, self.cfg.head_dim, self.cfg.num_kv_heads)
            rows = [[q0, min(q0 + blk, kv_end), 0, kv_end] for q0 in range(pos0, kv_end, blk)]
            rows = [rows[i] for i in ops.order_items(rows, blk, True)]      # 32x32 form: the blocks with the most key tiles first
            it = torch.tensor(rows, dtype=torch.int32).to(self.dev)
            it.q_block = blk
            self._item_cache[key] = it
        return it

    def prefill(self, embeds: torch.Tensor, pos: torch.Tensor, rope_delta: int = 0, collect: Optional[list] = None,
                tables: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """embeds [L, d] bf16 device, pos [3, L] host -> (final-norm last hidden [1, d], logits [1, V], next id tensor).
        `tables` = device (cos, sin) [L, head_dim] bf16 when the caller already uploaded them (graph replay)."""
        c = self.cfg
        self.reserve(embeds.shape[0])
        if tables is None:
            cos, sin = mrope_tables(pos, c.head_dim, c.rope_theta, c.mrope_section)
            cos, sin = cos.to(self.dev), sin.to(self.dev)
        else:
            cos, sin = tables
        with ops.workspace_scope(self._ws_owner):
            x = self._forward(embeds, cos, sin, 0, collect)
            self.kv_len = embeds.shape[0]
            self.rope_delta = rope_delta
            return self._head(x)

    # ---- several prompts packed into one pass (the reference's batch-aware splice, omchat_qwen2_5_vl.py:380-416) ----
    PACK_ALIGN = 4   # the attention kernel reads V^T in 8-byte (4-key) pieces: every sequence starts at a multiple of 4

    SHARE_MIN_ROWS = 256     # a common prefix shorter than this is not worth a second key range per attention item

    def plan_batch(self, prompts: Sequence[Sequence[int]], n_img: Sequence[int], n_regions: Sequence[int],
                   grids_merged: Sequence[Tuple[int, int]], img_base: Optional[Sequence[int]] = None, share_prefix: bool = False):
        """HOST: B prompts -> one packed plan.  Sequence b owns rows [off_b, off_b + Lp_b), Lp_b = its row count rounded up to PACK_ALIGN
        with dummy rows (token 0) appended AFTER its real rows: causal attention keeps them invisible to every real row.  Image /
        region indices address the batch-concatenated token tables.  Returns dict(plan int32 [R,2], cos, sin bf16 [R, hd] (host),
        seqs, pos [per-sequence [3, L]], delta [per sequence], last int32 [B,2] gather plan of the last real rows).

        seqs[b] = (off, L, Lp) — or, with share_prefix, (off, L, Lp, poff, P) for prompts over ONE image (same img_base) whose first P
        rows are identical (system text + the image block, up to the first region): those P rows (a multiple of PACK_ALIGN, at
        [poff, poff + P)) run through every layer ONCE, `off` / `Lp` describe the prompt's remaining L - P rows, which attend
        [prefix | own rows] (fo1_attention_prefix_bf16).  In a causal model the prefix rows' states do not depend on what follows, so
        every prompt's rows are what its own full pass computes (the reference would run the whole model once per prompt)."""
        c = self.cfg
        A = self.PACK_ALIGN
        B = len(prompts)
        per = []
        img0 = reg0 = 0
        for b, (ids, ni, nr, gm) in enumerate(zip(prompts, n_img, n_regions, grids_merged)):
            if img_base is not None:      # several prompts over ONE image: they all address that image's rows of the token table
                img0 = int(img_base[b])
            pl, pos, delta = self.plan_inputs(ids, ni, nr, gm)
            pl = pl.clone()
            pl[:, 1] += (pl[:, 0] == 1).to(torch.int32) * img0 + (pl[:, 0] == 2).to(torch.int32) * reg0
            per.append((pl, pos, delta))
            img0 += ni; reg0 += nr
        # prompts that share a prefix: same image rows AND the same plan rows up to the first difference
        shared = {}                       # first member -> (P, members)
        member_of = [None] * B
        if share_prefix and img_base is not None:
            by_img = {}
            for b in range(B):
                by_img.setdefault(int(img_base[b]), []).append(b)
            for members in by_img.values():
                if len(members) < 2:
                    continue
                p0 = per[members[0]][0]
                P = p0.shape[0]
                for m in members[1:]:
                    pm = per[m][0]
                    n = min(P, pm.shape[0])
                    diff = (p0[:n] != pm[:n]).any(dim=1).nonzero()
                    P = int(diff[0]) if diff.numel() else n
                P = min(P, min(per[m][0].shape[0] for m in members) - 1) // A * A      # every prompt keeps >= 1 own row (its last row feeds the head)
                if P >= self.SHARE_MIN_ROWS:
                    shared[members[0]] = (P, members)
                    for m in members:
                        member_of[m] = members[0]
        plans, coss, sins, seqs, poss, deltas, last_rows = [], [], [], [None] * B, [], [], []
        prefix_at = {}
        off = 0
        for b in range(B):
            pl, pos, delta = per[b]
            L = pl.shape[0]
            P = shared[member_of[b]][0] if member_of[b] is not None else 0
            own = L - P
            Lp = (own + A - 1) // A * A
            if Lp > own:
                pl = torch.cat([pl, torch.zeros(Lp - own, 2, dtype=torch.int32)], 0)
                tail = int(pos.max()) + 1 + torch.arange(Lp - own).view(1, -1).expand(3, -1)
                pos_p = torch.cat([pos, tail], 1)
            else:
                pos_p = pos
            cs, sn = mrope_tables(pos_p, c.head_dim, c.rope_theta, c.mrope_section)
            if P and member_of[b] == b:   # the group's first prompt: its prefix rows are the group's
                plans.append(pl[:P]); coss.append(cs[:P]); sins.append(sn[:P])
                prefix_at[b] = off
                off += P
            plans.append(pl[P:]); coss.append(cs[P:]); sins.append(sn[P:])
            seqs[b] = (off, L, Lp) if not P else (off, L, Lp, prefix_at[member_of[b]], P)
            last_rows.append(off + own - 1)
            poss.append(pos); deltas.append(delta)
            off += Lp
        last = torch.tensor([[0, r] for r in last_rows], dtype=torch.int32).reshape(-1, 2)
        return dict(plan=torch.cat(plans, 0).contiguous(), cos=torch.cat(coss, 0).contiguous(), sin=torch.cat(sins, 0).contiguous(),
                    seqs=seqs, pos=poss, delta=deltas, last=last, rows=off)

    def packed_items(self, seqs):
        """-> (attention work items int32 [n, 4] on the device, flop count, prefix ranges int32 [n, 2] or None)."""
        key = ("packed", tuple(seqs))
        hit = self._item_cache.get(key)
        if hit is None:
            segs = []                                  # (q0, q1, second range) — one per sequence, one per distinct shared prefix
            done = set()
            fl = 0.0
            for sq in seqs:
                o, _, Lp, *pre = sq
                if pre:
                    po, P = pre
                    if (po, P) not in done:
                        done.add((po, P))
                        segs.append((po, po + P, (0, 0)))
                        fl += P * (P + 1) / 2.0
                    segs.append((o, o + Lp, (po, po + P)))
                    fl += Lp * (Lp + 1) / 2.0 + float(Lp) * P
                else:
                    segs.append((o, o + Lp, (0, 0)))
                    fl += Lp * (Lp + 1) / 2.0
            blk = ops.pick_q_block([(a, b) for a, b, _ in segs], self.cfg.num_heads, self.cfg.head_dim, self.cfg.num_kv_heads)
            rows = [[q0, min(q0 + blk, b), a, b] for a, b, _ in segs for q0 in range(a, b, blk)]
            rng = [list(r2) for a, b, r2 in segs for _ in range(a, b, blk)]
            order = ops.order_items(rows, blk, True, prefix=rng)           # 32x32 form: longest-processing-time-first over the launch
            rows, rng = [rows[i] for i in order], [rng[i] for i in order]
            it = torch.tensor(rows, dtype=torch.int32).to(self.dev)
            it.q_block = blk
            r2 = torch.tensor(rng, dtype=torch.int32).to(self.dev) if any(x[1] > x[0] for x in rng) else None
            hit = (it, 4.0 * self.cfg.num_heads * self.cfg.head_dim * fl, r2)
            self._item_cache[key] = hit
        return hit

    def prefill_packed(self, embeds: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, seqs, last_plan: torch.Tensor,
                       collect: Optional[list] = None):
        """DEVICE: embeds [R, d] (packed rows of plan_batch), cos/sin [R, hd] device tables, last_plan int32 [B,2] (device).
        K / V^T land at cache positions = packed row indices.  Returns (final-norm last hidden [B, d], logits [B, V], next ids
        int32 [B])."""
        c = self.cfg
        items, flops, prefix_ranges = self.packed_items(seqs)
        from . import stage_abi
        if stage_abi.enabled() and collect is None and prefix_ranges is None:      # the same launches, sequenced by fo1_llm_prefill (csrc/stages.hip)
            if embeds.shape[0] > self.capacity:
                raise ValueError(f"sequence {embeds.shape[0]} exceeds the KV cache ({self.capacity}); call reserve() first")
            with ops.workspace_scope(self._ws_owner):
                return stage_abi.llm_stage(self).prefill_packed(embeds, cos, sin, seqs, last_plan)
        with ops.workspace_scope(self._ws_owner):
            x = self._forward(embeds, cos, sin, 0, collect, items=items, flops=flops, prefix_ranges=prefix_ranges)
            last = ops.rmsnorm(ops.gather_rows(last_plan, c.hidden_size, x), self.norm, c.rms_norm_eps)
            logits = ops.gemm(last, self.lm_head)
            toks = torch.empty(last.shape[0], dtype=torch.int32, device=self.dev)
            for b in range(last.shape[0]):
                ops.argmax(logits[b], out=toks[b:b + 1])
            return last, logits, toks

    def _check_room(self):
        """The device-side decode state indexes the caches blindly: grow them on the host before a step would run past them."""
        if self.kv_len + 1 > self.capacity:
            self.reserve(self.kv_len + 1)
            self.sync_decode_state()
        if self.rope_cos.shape[0] < self.capacity:      # the step reads table row (position + rope delta) < capacity
            self.grow_rope(self.capacity)

    def sync_decode_state(self):
        """Publish (kv_len, rope row, first decode work item) to the device-side decode state."""
        n = self.kv_len
        st = torch.tensor([n, n + self.rope_delta, 0, 0, n, n + 1, 0, n + 1], dtype=torch.int32)
        self.dstate.copy_(st, non_blocking=True)
        self._dstate_keep = st      # keep the source alive until the next upload replaces it

    def _head(self, x: torch.Tensor):
        c = self.cfg
        last = ops.rmsnorm(x[-1:], self.norm, c.rms_norm_eps)
        logits = ops.gemm(last, self.lm_head)  # last row only: output-identical to the reference's all-row lm_head
        tok = ops.argmax(logits[0])
        return last, logits, tok

    def _decode_device(self):
        """One decode step whose every position-dependent quantity is read from device memory (self.dstate /
        self.dplan): capturable once, replayable for every token."""
        with ops.workspace_scope(self._ws_owner):
            return self._decode_device_impl()

    def _decode_device_impl(self):
        c = self.cfg
        H, KV, HD = c.num_heads, c.num_kv_heads, c.head_dim
        x = ops.gather_rows(self.dplan, c.hidden_size, self.embed)
        scale = 1.0 / math.sqrt(HD)
        pos_ptr = self.dstate[0:1]
        items = self.dstate[4:8].view(1, 4)
        for li, w in enumerate(self.layers):
            # 7 launches per layer: norms folded into the GEMV prologues, rope + K/V^T cache append in one kernel
            qkv = ops.gemv(x, w["wqkv"], w["bqkv"], norm_weight=w["ln1"], norm_eps=c.rms_norm_eps)
            ops.decode_qkv_post(qkv, H, KV, HD, self.rope_cos, self.rope_sin, self.dstate, self.kcache[li], self.vtcache[li])
            # split-KV decode attention: chunk count follows the device-side kv length (= dstate[7] = position + 1)
            att = ops.attention_decode(qkv[:, :H * HD], self.kcache[li], self.vtcache[li], self.dstate[7:8], self.capacity, H, KV, HD, scale)
            x = ops.gemv(att, w["wo"], residual=x)
            a = ops.gemv(x, w["wgu"], act=ops.ACT_SWIGLU16, norm_weight=w["ln2"], norm_eps=c.rms_norm_eps)
            x = ops.gemv(a, w["wdown"], residual=x)
        logits = ops.gemv(x, self.lm_head, norm_weight=self.norm, norm_eps=c.rms_norm_eps)
        ops.argmax(logits[0], out=self.dplan.view(-1)[1:2])    # next token id lands in the gather plan of the next step
        ops.decode_advance(self.dstate)
        return logits

    def decode_step_graph(self, token_id: Optional[torch.Tensor] = None):
        """Greedy step via one replayed hipGraph.  `token_id` (device int32 [1]) seeds the plan on the first step
        after a prefill; later steps reuse the id the previous replay wrote.  Returns (logits, next id tensor)."""
        self._check_room()
        if token_id is not None:
            self.dplan.view(-1)[1:2].copy_(token_id.to(torch.int32).view(1), non_blocking=True)
        if self._dgraph is None:
            with ops.graph_lock.capture(), torch.inference_mode(False):   # exclusive (see ops._CaptureLock); mode: see FO1Engine._capture
                snap = self.dstate.clone()
                plan = self.dplan.clone()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._decode_device()          # warm-up: allocates scratch (the step it computes is re-done below)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.dstate.copy_(snap)
                self.dplan.copy_(plan)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL watchdog threads may touch the runtime meanwhile
                    logits = self._decode_device()
                self.dstate.copy_(snap)            # the capture itself does not execute, but keep the state exact
                self.dplan.copy_(plan)
                self._dgraph = (g, logits)
        g, logits = self._dgraph
        with ops.graph_lock.replay():
            g.replay()
        self.kv_len += 1
        return logits, self.dplan.view(-1)[1:2]

    def decode_step(self, token_id: torch.Tensor):
        """One greedy step with individually launched kernels (same device code as the graph path, so the two are
        bit-identical): token_id int32 [1] on device -> (last hidden placeholder, logits [1, V], next id tensor).
        Position = cache_position + rope_delta on all three axes (reference :1848-1860)."""
        self._check_room()
        self.dplan.view(-1)[1:2].copy_(token_id.to(torch.int32).view(1), non_blocking=True)
        self.sync_decode_state()
        logits = self._decode_device()
        self.kv_len += 1
        return None, logits, self.dplan.view(-1)[1:2].clone()


def reloc_rows(seqs, dst0s):
    """kv_relocate entries [src0, dst0, len, 0] that move every sequence of a packed prefill to cache row dst0: one entry, or two for a
    sequence whose first P rows are a prefix shared with other prompts (plan_batch(share_prefix=True): (off, L, Lp, poff, P))."""
    rows = []
    for sq, d in zip(seqs, dst0s):
        o, L, _, *pre = sq
        if pre:
            po, P = pre
            rows += [[po, d, P, 0], [o, d + P, L - P, 0]]
        else:
            rows.append([o, d, L, 0])
    return rows


class BatchDecoder:
    """Greedy decode of up to 32 sequences at once: the weights are streamed ONCE per step for all sequences
    (fo1_gemv_batch_bf16), 5 launches per layer, and the whole step — embedding gather, 36 layers, lm_head, argmax, stop check,
    position bookkeeping — replays as one hipGraph with no host read inside the loop (the host polls a device-side `done` counter
    every few steps).  Reference semantics: HF greedy search over the 1-token fast path of omchat_qwen2_5_vl.py:143-155;
    positions = cache position + rope delta (modeling_qwen2_5_vl.py:1848-1860); a sequence stops AFTER its EOS / keyword id has
    been appended, or at max_new_tokens (mm_utils.py:137-181, 640-654)."""
    MAX_BATCH = 32          # 17..32 sequences: two 16-column groups per weight fragment (decode_mfma.hip, MM = 32)
    FUSED_COMBINE_MAX = 2   # up to this many sequences the o-projection combines the split-KV attention partials itself (0 = always the combine launch; A/B)
    IDS_CAP = 4096          # generated ids kept per sequence (every reference caller uses max_new_tokens <= 4096)
    MAX_STOP = 16           # stop ids the device-side rule compares against

    def __init__(self, llm: QwenLLM):
        self.llm = llm
        dev = llm.dev
        B = self.MAX_BATCH
        # persistent buffers are updated in place for the life of the engine: ordinary tensors even when the first request arrives
        # under the caller's torch.inference_mode() (the reference's inference.py:46)
        with torch.inference_mode(False):
            self.state = torch.zeros(B, 8, dtype=torch.int32, device=dev)
            self.plan = torch.zeros(B, 2, dtype=torch.int32, device=dev)
            self.ids = torch.zeros(B, self.IDS_CAP, dtype=torch.int32, device=dev)
            self.done = torch.zeros(1, dtype=torch.int32, device=dev)
            self.stop = torch.zeros(self.MAX_STOP, dtype=torch.int32, device=dev)
            self.reloc = torch.zeros(2 * B, 4, dtype=torch.int32, device=dev)      # (a shared-prefix sequence moves in two pieces)
        self.n_stop = 0
        self.dk = self.dvt = None
        self.rows = 0
        self.slot = 0
        self.B = 0
        self._graphs: Dict[tuple, tuple] = {}
        self._keep: list = []
        self._ws_owner = ops.new_owner(self)

    def _ensure(self, rows: int, slot: int):
        c = self.llm.cfg
        if rows > self.rows:
            bf = torch.bfloat16
            with torch.inference_mode(False):
                self.dk = torch.zeros(c.num_layers, c.num_kv_heads, rows, c.head_dim, dtype=bf, device=self.llm.dev)
                self.dvt = torch.zeros(c.num_layers, c.num_kv_heads * c.head_dim, rows, dtype=bf, device=self.llm.dev)
            self.rows = rows
            self._graphs = {}
        # The rope table is indexed by POSITION (state[1] = L + rope delta + steps < slot), never by cache row: it needs `slot` rows,
        # not B * slot.  When it does grow, every graph that baked the old pointers in goes: this decoder's and the
        # single-sequence decode graph of the llm (the streamer path would otherwise replay against freed memory).
        if self.llm.rope_cos.shape[0] < slot:
            self.llm.grow_rope(slot)
            self._graphs = {}

    def start(self, seqs, deltas, first_tokens: torch.Tensor, max_new_tokens: int, stop_ids: Sequence[int] = ()):
        """seqs: [(cache row offset, L, ...)] of the packed prefill that just ran on llm.kcache / llm.vtcache; first_tokens: device
        int32 [B] (the prefill's greedy picks).  Moves every sequence to its own slot and accepts the first tokens on the device."""
        llm = self.llm
        B = len(seqs)
        if B > self.MAX_BATCH:
            raise ValueError(f"BatchDecoder handles at most {self.MAX_BATCH} sequences")
        # no silent truncation: the HF-like contract is "exactly these stop ids, exactly this budget"
        stop_ids = sorted(set(int(t) for t in stop_ids))
        if len(stop_ids) > self.MAX_STOP:
            raise ValueError(f"BatchDecoder evaluates at most {self.MAX_STOP} stop ids on the device (got {len(stop_ids)}); use the host loop")
        if int(max_new_tokens) > self.IDS_CAP:
            raise ValueError(f"BatchDecoder keeps at most {self.IDS_CAP} generated ids per sequence (max_new_tokens={int(max_new_tokens)}); use the host loop")
        max_new = max(1, int(max_new_tokens))
        need = max(L for _, L, *_ in seqs) + max_new + 1
        slot = 1024
        while slot < need:
            slot *= 2
        self._ensure(max(B * slot, self.rows), slot)
        rope_id = (llm.rope_epoch, llm.rope_cos.data_ptr())
        if rope_id != getattr(self, "_rope_id", rope_id):
            self._graphs = {}              # graphs of the previous table are unreachable by key: release them and their pools
        self._rope_id = rope_id
        self.B, self.slot = B, slot
        self.len0, self.steps = max(L for _, L, *_ in seqs), 0      # host-side bound on any sequence's keys: len0 + steps + 1
        reloc = torch.tensor(reloc_rows(seqs, [b * slot for b in range(B)]), dtype=torch.int32)
        nr = reloc.shape[0]
        state = torch.tensor([[b * slot + L, L + d, b * slot, 0, 0, max_new, 0, 0] for b, ((o, L, *_), d) in enumerate(zip(seqs, deltas))],
                             dtype=torch.int32)
        stop = torch.tensor(stop_ids + [0] * (self.MAX_STOP - len(stop_ids)), dtype=torch.int32)
        self.n_stop = len(stop_ids)
        self.reloc[:nr].copy_(reloc, non_blocking=True)
        self.state[:B].copy_(state, non_blocking=True)
        self.stop.copy_(stop, non_blocking=True)
        self._keep = [reloc, state, stop]                 # sources of the async uploads stay alive
        self.done.zero_()
        with ops.workspace_scope(self._ws_owner):
            ops.kv_relocate(llm.kcache, self.dk, llm.vtcache, self.dvt, self.reloc[:nr], max(L for _, L, *_ in seqs))
            ops.decode_argmax_accept(None, first_tokens.to(torch.int32).contiguous(), self.state[:B], self.plan[:B], self.ids[:B],
                                     self.stop[:self.n_stop], self.done)

    KV_BUCKET = 2048

    def kv_bucket(self) -> int:
        """Upper bound on the keys any sequence attends in the coming step, rounded up to KV_BUCKET rows (capped by the slot): the
        attention launch geometry follows it (one workgroup per KV head and sequence up to 2048 keys), so a 4096-token budget does
        not make every step pay the long-context split.  A new bucket means a new captured graph, once per 2048 generated tokens."""
        need = self.len0 + self.steps + 1
        return min(self.slot, -(-need // self.KV_BUCKET) * self.KV_BUCKET)

    def _step_device(self):
        llm, B = self.llm, self.B
        c = llm.cfg
        H, KV, HD = c.num_heads, c.num_kv_heads, c.head_dim
        scale = 1.0 / math.sqrt(HD)
        st = self.state[:B]
        from . import stage_abi
        if stage_abi.enabled():      # the same launches, sequenced by fo1_llm_decode_step (csrc/stages.hip)
            with ops.workspace_scope(self._ws_owner):
                return stage_abi.llm_stage(llm).decode_step(self)
        with ops.workspace_scope(self._ws_owner):
            x = ops.gather_rows(self.plan[:B], c.hidden_size, llm.embed)
            for li, w in enumerate(llm.layers):
                q = ops.gemv_batch(x, w["wqkv"], w["bqkv"], mode=ops.GB_QKV, norm_weight=w["ln1"], norm_eps=c.rms_norm_eps,
                                   qkv=dict(n_q=H, n_kv=KV, cos=llm.rope_cos, sin=llm.rope_sin, state=st, kcache=self.dk[li], vtcache=self.dvt[li]))
                if B <= self.FUSED_COMBINE_MAX and H * HD <= 2048 and c.hidden_size <= 4096:
                    # one or two sequences (the reference's own batch-1 loop): the o-projection sums the split-KV partials in its prologue — same
                    # bits as the combine launch (one shared routine, csrc/decode_common.h), one launch less per layer
                    part, pstride, chunk = ops.attention_decode_batch_partials(q, self.dk[li], self.dvt[li], st, self.kv_bucket(), H, KV, HD, scale)
                    x = ops.gemv_attn_combine(part, pstride, st, chunk, H, KV, w["wo"], residual=x)
                else:
                    att = ops.attention_decode_batch(q, self.dk[li], self.dvt[li], st, self.kv_bucket(), H, KV, HD, scale)
                    x = ops.gemv_batch(att, w["wo"], residual=x)
                a = ops.gemv_batch(x, w["wgu"], mode=ops.GB_SWIGLU, norm_weight=w["ln2"], norm_eps=c.rms_norm_eps)
                x = ops.gemv_batch(a, w["wdown"], residual=x)
            logits = ops.gemv_batch(x, llm.lm_head, norm_weight=llm.norm, norm_eps=c.rms_norm_eps)
            ops.decode_argmax_accept(logits, None, st, self.plan[:B], self.ids[:B], self.stop[:self.n_stop], self.done)
            return logits

    def step(self, use_graph: bool = True):
        """One token for every live sequence."""
        if not use_graph:
            out = self._step_device()
            self.steps += 1
            return out
        # the rope tables' identity is part of the key: QwenLLM.reserve() on a larger later batch or a sibling decoder may re-allocate
        # them (grow_rope) while this decoder's graphs survive — a replay would read the freed table
        key = (self.B, self.slot, self.n_stop, self.kv_bucket(), self.llm.rope_epoch, self.llm.rope_cos.data_ptr())
        ent = self._graphs.get(key)
        if ent is None:
            with ops.graph_lock.capture(), torch.inference_mode(False):
                snap = (self.state.clone(), self.plan.clone(), self.ids.clone(), self.done.clone())
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._step_device()          # warm-up (allocates scratch); its effects are rolled back below
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                for dst, src in zip((self.state, self.plan, self.ids, self.done), snap):
                    dst.copy_(src)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    logits = self._step_device()
                ent = (g, logits)
                self._graphs[key] = ent
        with ops.graph_lock.replay():
            ent[0].replay()
        self.steps += 1
        return ent[1]

    def results(self) -> List[List[int]]:
        """Generated ids per sequence so far (first token included)."""
        B = self.B
        n = self.state[:B, 4].cpu().tolist()
        ids = self.ids[:B, :max(n)].cpu().tolist()
        return [row[:k] for row, k in zip(ids, n)]

    def run(self, max_new_tokens: int, use_graph: bool = True, poll: int = 8) -> List[List[int]]:
        """Decode until every sequence has stopped (or max_new_tokens).  Returns the generated ids per sequence, the first
        token (from the prefill) included."""
        B = self.B
        max_new = max(1, min(int(max_new_tokens), self.IDS_CAP))      # start() has already refused a larger budget
        for i in range(max_new - 1):
            if i % poll == 0 and int(self.done.item()) >= B:     # the only host read in the loop, every `poll` steps
                break
            self.step(use_graph)
        n = self.state[:B, 4].cpu().tolist()
        ids = self.ids[:B, :max(n)].cpu().tolist()
        return [row[:k] for row, k in zip(ids, n)]


def run_decoders(decs: Sequence["BatchDecoder"], streams: Sequence, max_new_tokens: int, use_graph: bool = True, poll: int = 8) -> List[List[List[int]]]:
    """Several started decode groups advanced TOGETHER, each on its own HIP stream: a step of one group is a chain of ~180 short
    kernels (5 per layer) with launch gaps between them and ~2.5 TB/s of an 8 TB/s memory system in use — the chains of two groups
    interleave on the GPU, so a pass of 25 sequences (13 + 12) finishes its 64 tokens in about the time one group of 16 takes alone
    instead of the sum of two (bench.py `end_to_end`).  Per sequence nothing changes: same kernels, same sums, same ids.
    -> [ids per sequence] per group."""
    max_new = max(1, min(int(max_new_tokens), BatchDecoder.IDS_CAP))
    live = list(range(len(decs)))
    for i in range(max_new - 1):
        if i % poll == 0:      # the only host reads: each group's `done` counter on the group's own stream, every `poll` steps
            still = []
            for k in live:
                with torch.cuda.stream(streams[k]):
                    if int(decs[k].done.item()) < decs[k].B:
                        still.append(k)
            live = still
            if not live:
                break
        for k in live:
            with torch.cuda.stream(streams[k]):
                decs[k].step(use_graph)
    out = []
    for k, d in enumerate(decs):
        with torch.cuda.stream(streams[k]):
            out.append(d.results())
    return out


class DecodePool:
    """Continuous batching for the greedy decode loop: a pool of 64 / 128 sequence SLOTS that advance together,
    one token per step, through ONE stream of the weights.  Sequences JOIN whenever a packed prefill pass has finished (their K / V^T
    rows are relocated out of that pass's cache into free slots, first tokens accepted on the device) and LEAVE individually when their
    stop rule fires (device-side, as in BatchDecoder); the step itself always computes all P rows — empty and finished slots are rows
    nobody reads, their attention is skipped on the device.  Every position-dependent quantity lives in device memory, so one captured
    hipGraph per (kv-length bucket) serves every step whatever the occupancy.

    Past 32 sequences the step's projections are skinny GEMMs C[P, N] = x[P, K] W[N, K]^T: the prefill's LDS-DMA tile kernels at M = P
    (fo1_gemm_bf16: 64 x 64 / 64 x 128 rings for the few-row shapes, 128 x 128 for gate/up and lm_head), RMSNorm and the q/k/v
    post-processing (mRoPE at each slot's position + cache append, fo1_pool_qkv_post_bf16) as their own launches, the split-KV decode
    attention with 256 keys per workgroup.  A dedicated weights-to-VGPR streaming kernel was built and measured first
    (profiles/r04_pool_gemm_stream_kernel_vs_tile_kernels.json: 1.3-1.6x SLOWER than the tile kernels on every shape — two K tiles of
    register prefetch cannot cover the HBM latency that a 3-4 stage LDS ring does) and is not in the tree.

    Reference semantics per sequence: HF greedy search over the 1-token fast path of omchat_qwen2_5_vl.py:143-155; positions = cache
    position + rope delta (modeling_qwen2_5_vl.py:1848-1860); stop AFTER the EOS / keyword id has been appended, or at max_new_tokens
    (mm_utils.py:137-181, 640-654).  A sequence's ids do not depend on its slot or on its neighbours (every kernel's per-row fp32 sum
    order is a function of the launch shape only, and the shape is always P rows; tests/test_decode_pool_gpu.py)."""
    IDS_CAP = 4096
    MAX_STOP = 16
    MAX_SETS = 32            # distinct stop-id sets live in the pool at once
    KV_BUCKET = 256          # the attention launch geometry follows the longest live context rounded up to this many keys

    def __init__(self, llm: QwenLLM, slots: int = 128, slot_rows: int = 1024):
        if slots not in (64, 128):
            raise ValueError("a decode pool has 64 or 128 slots")
        self.llm, self.P = llm, slots
        dev = llm.dev
        P = slots
        with torch.inference_mode(False):
            st = torch.zeros(P, 8, dtype=torch.int32)
            st[:, 3] = 1                               # every slot starts empty = "finished": its attention is skipped, its ids ignored
            self.state = st.to(dev)
            self.plan = torch.zeros(P, 2, dtype=torch.int32, device=dev)
            self.ids = torch.zeros(P, self.IDS_CAP, dtype=torch.int32, device=dev)
            self.done = torch.zeros(1, dtype=torch.int32, device=dev)
            # per-sequence stop rules (round 5): a table of stop-id sets {count, ids[16]}; state[slot][6] names a slot's set, so submissions
            # with different stop ids / templates share the pool (before: one set per pool, anything else waited for a full drain)
            self.stop = torch.zeros(self.MAX_SETS, 1 + self.MAX_STOP, dtype=torch.int32, device=dev)
            self.reloc = torch.zeros(2 * P, 4, dtype=torch.int32, device=dev)     # (a shared-prefix sequence moves in two pieces)
        self._sets: Dict[tuple, int] = {}              # stop-id tuple -> row of self.stop
        self._set_users = [0] * self.MAX_SETS          # live slots per row (a row with users is never rewritten)
        self.slot_set = [0] * P
        self.slot_rows = 0
        self.dk = self.dvt = None
        self.free = list(range(P))                     # host view of the slots
        self.bound = [0] * P                           # upper bound of the keys a live slot attends in the coming step
        self.budget = [0] * P
        self.live: Dict[int, object] = {}              # slot -> caller's tag
        self.steps_run = 0
        self._graphs: Dict[tuple, tuple] = {}
        self._keep: list = []
        self._ws_owner = ops.new_owner(self)
        self._ensure(slot_rows)
        self._tiled()

    # ---- memory ----------------------------------------------------------------------------------------------------------------
    def _ensure(self, slot_rows: int):
        """Slots of at least slot_rows cache rows each (power of two).  Re-allocation is only possible while the pool is empty."""
        rows = 1024
        while rows < slot_rows:
            rows *= 2
        if rows > self.slot_rows:
            if self.live:
                raise RuntimeError("DecodePool: slots can only grow while the pool is empty")
            c = self.llm.cfg
            bf = torch.bfloat16
            with torch.inference_mode(False):
                self.dk = self.dvt = None
                self.dk = torch.zeros(c.num_layers, c.num_kv_heads, self.P * rows, c.head_dim, dtype=bf, device=self.llm.dev)
                self.dvt = torch.zeros(c.num_layers, c.num_kv_heads * c.head_dim, self.P * rows, dtype=bf, device=self.llm.dev)
            self.slot_rows = rows
            self._graphs = {}
            # an empty / finished slot still runs every step: its (ignored) K / V^T rows must land inside ITS OWN slot — row 0 of the
            # cache is slot 0's first prompt token
            st = torch.zeros(self.P, 8, dtype=torch.int32)
            st[:, 0] = st[:, 2] = torch.arange(self.P, dtype=torch.int32) * rows
            st[:, 3] = 1
            with torch.inference_mode(False):
                self.state.copy_(st)
        if self.llm.rope_cos.shape[0] < self.slot_rows:
            self.llm.grow_rope(self.slot_rows)
            self._graphs = {}

    def fits(self, seqs, max_new_tokens: int) -> bool:
        return max(L for _, L, *_ in seqs) + max(1, int(max_new_tokens)) + 1 <= self.slot_rows

    # ---- join ------------------------------------------------------------------------------------------------------------------
    def join(self, kcache: torch.Tensor, vtcache: torch.Tensor, seqs, deltas, first_tokens: torch.Tensor, max_new_tokens: int,
             stop_ids: Sequence[int] = (), tags: Optional[Sequence] = None) -> List[int]:
        """seqs [(cache row offset, L, ...)] of a packed prefill that ran on (kcache, vtcache); first_tokens device int32 [B].  Enqueues,
        on the CURRENT stream (the pool's), the relocation of every sequence into a free slot and the on-device accept of its first
        token.  Returns the slots.  Raises if there is no room (callers check `len(pool.free)` / `fits()` first)."""
        B = len(seqs)
        stop_ids = tuple(sorted(set(int(t) for t in stop_ids)))
        if len(stop_ids) > self.MAX_STOP:
            raise ValueError(f"DecodePool evaluates at most {self.MAX_STOP} stop ids on the device (got {len(stop_ids)})")
        set_row = self._stop_set_row(stop_ids)
        max_new = max(1, int(max_new_tokens))
        if max_new > self.IDS_CAP:
            raise ValueError(f"DecodePool keeps at most {self.IDS_CAP} generated ids per sequence (max_new_tokens={max_new})")
        if B > len(self.free):
            raise RuntimeError(f"DecodePool: {B} sequences, {len(self.free)} free slots")
        if not self.fits(seqs, max_new):
            self._ensure(max(L for _, L, *_ in seqs) + max_new + 1)      # raises unless the pool is empty
        self.free.sort()
        slots = [self.free.pop(0) for _ in range(B)]
        try:
            self._join_slots(slots, kcache, vtcache, seqs, deltas, first_tokens, max_new, set_row)
        except BaseException:
            # nothing of this submission is live: its slots go back (their device state may be half written — the next occupant's
            # join rewrites state, plan and K / V^T rows; until then the slot is marked finished so that no step reads it)
            done = torch.zeros(len(slots), 8, dtype=torch.int32)
            done[:, 0] = done[:, 2] = torch.tensor(slots, dtype=torch.int32) * self.slot_rows
            done[:, 3] = 1
            self._keep.append(done)
            try:
                for k, sl in enumerate(slots):
                    self.state[sl:sl + 1].copy_(done[k:k + 1], non_blocking=True)
            except BaseException:
                pass
            self.free += slots
            raise
        for k, s in enumerate(slots):
            # a tag identifies ONE occupancy of a slot: harvest() compares it by identity against an older snapshot, so the default must
            # be unique per join (with None, `None is None` handed a re-joined slot's new occupant the old occupant's ids — ADVICE r4)
            self.live[s] = tags[k] if tags is not None else object()
            self.bound[s] = seqs[k][1] + 1
            self.budget[s] = max_new
            self.slot_set[s] = set_row
            self._set_users[set_row] += 1
        return slots

    def can_take(self, stop_ids: Sequence[int]) -> bool:
        """Is there a table row for this stop-id set (its own, or one no live slot uses)?"""
        key = tuple(sorted(set(int(t) for t in stop_ids)))
        return key in self._sets or any(u == 0 for u in self._set_users)

    def _stop_set_row(self, stop_ids: tuple) -> int:
        row = self._sets.get(stop_ids)
        if row is None:
            free = [r for r in range(self.MAX_SETS) if self._set_users[r] == 0 and r not in self._sets.values()] or \
                   [r for r in range(self.MAX_SETS) if self._set_users[r] == 0]
            if not free:
                raise RuntimeError(f"DecodePool: {self.MAX_SETS} different stop-id sets are live; wait for sequences to finish")
            row = free[0]
            for k in [k for k, v in self._sets.items() if v == row]:
                del self._sets[k]
            sv = torch.tensor([[len(stop_ids)] + list(stop_ids) + [0] * (self.MAX_STOP - len(stop_ids))], dtype=torch.int32)
            self.stop[row:row + 1].copy_(sv, non_blocking=True)
            self._keep.append(sv)
            self._sets[stop_ids] = row
        return row

    def _join_slots(self, slots, kcache, vtcache, seqs, deltas, first_tokens, max_new, set_row):
        B = len(slots)
        R = self.slot_rows
        state = torch.tensor([[s * R + L, L + d, s * R, 0, 0, max_new, set_row, 0] for s, (_, L, *_), d in zip(slots, seqs, deltas)], dtype=torch.int32)
        first = first_tokens.to(torch.int32).contiguous()
        self._keep += [state]
        if len(self._keep) > 64:
            del self._keep[:len(self._keep) - 64]
        # contiguous runs of slots: one relocate + one accept launch per run
        i = 0
        with ops.workspace_scope(self._ws_owner):
            while i < B:
                j = i
                while j + 1 < B and slots[j + 1] == slots[j] + 1:
                    j += 1
                a, n = slots[i], j - i + 1
                reloc = torch.tensor(reloc_rows(seqs[i:j + 1], [s * R for s in slots[i:j + 1]]), dtype=torch.int32)
                nr = reloc.shape[0]                      # <= 2 n: the run's entries live at rows [2 a, 2 a + nr) of the device table
                self._keep.append(reloc)
                self.reloc[2 * a:2 * a + nr].copy_(reloc, non_blocking=True)
                self.state[a:a + n].copy_(state[i:j + 1], non_blocking=True)
                ops.kv_relocate(kcache, self.dk, vtcache, self.dvt, self.reloc[2 * a:2 * a + nr], max(L for _, L, *_ in seqs[i:j + 1]))
                ops.decode_argmax_accept(None, first[i:j + 1], self.state[a:a + n], self.plan[a:a + n], self.ids[a:a + n],
                                         self.stop, self.done, per_sequence_sets=True)
                i = j + 1

    # ---- step ------------------------------------------------------------------------------------------------------------------
    def kv_bucket(self) -> int:
        need = max((self.bound[s] for s in self.live), default=1)
        return min(self.slot_rows, -(-need // self.KV_BUCKET) * self.KV_BUCKET)

    # Split-K planes + fused consumers for the few-tile projections (q/k/v: 80 output tiles, o: 64, down: 32 on 256 CUs): the GEMM writes
    # fp32 planes from >= 256 workgroups and the kernel that needs the result anyway sums them — q/k/v in the RoPE / cache-append kernel, o and
    # down in ONE kernel with the residual add and the NEXT RMSNorm (9 launches per layer instead of 11, none of them on a quarter of the chip).
    FUSED_SPLITK = True
    # gateup 0 = one GEMM with the SwiGLU epilogue (172 tiles of 128 x 128: 34 us); as 2 / 3 / 4 planes of 128 x 256 tiles + fo1_splitk_swiglu_bf16 it
    # measured 36 / 49 / 46 us (profiles/r04_pool_step_splitk_sweep.json) — the planes path stays available, parity-tested, unused
    SPLITS = dict(qkv=4, o=4, down=8, gateup=0)
    # TILED_WEIGHTS: gate/up and lm_head read a copy pre-tiled as [N / 128][K / 64][128][64] (ops.tile_weight, fo1_gemm_bf16_wtiled: a K tile is one
    # contiguous 16 KB block; 3.8 GB at the 3B shapes).  Bit-identical, and the weight stream ALONE gains 15 % that way (profiles/r04_hbm_stream_patterns.jsonl),
    # but the step does not: 3.872 vs 3.888 ms (profiles/r04_pool_step_tiled_weights_ab.json) — the activations on the same DMA path are what the tile waits for.  Off.
    TILED_WEIGHTS = False

    def _tiled(self):
        """(per-layer tiled gate/up copies, tiled lm_head) or None; built once per LLM, in DecodePool.__init__ (never inside a graph capture)."""
        if not self.TILED_WEIGHTS:
            return None
        t = self.llm.__dict__.get("_pool_tiled_w")          # on the LLM: every pool over these weights shares the copies
        if t is None:
            def tile(w):
                return ops.tile_weight(w) if (w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0) else None
            with torch.inference_mode(False):
                t = ([tile(w["wgu"]) for w in self.llm.layers], tile(self.llm.lm_head))
            self.llm._pool_tiled_w = t
        return t

    def _step_device(self, bucket: int):
        llm, P = self.llm, self.P
        c = llm.cfg
        H, KV, HD = c.num_heads, c.num_kv_heads, c.head_dim
        scale = 1.0 / math.sqrt(HD)
        st = self.state
        with ops.workspace_scope(self._ws_owner):
            x = ops.gather_rows(self.plan, c.hidden_size, llm.embed)
            if self.FUSED_SPLITK:
                eps, D = c.rms_norm_eps, c.hidden_size
                NGU = llm.layers[0]["wgu"].shape[0]
                part = torch.empty(max(self.SPLITS["qkv"] * (H + 2 * KV) * HD, max(self.SPLITS["o"], self.SPLITS["down"]) * D, self.SPLITS["gateup"] * NGU) * P, dtype=torch.float32,
                                   device=x.device)
                xn = ops.rmsnorm(x, llm.layers[0]["ln1"], eps)
                q = torch.empty(P, H * HD, dtype=torch.bfloat16, device=x.device)
                for li, w in enumerate(llm.layers):
                    s = ops.gemm_partials(xn, w["wqkv"], self.SPLITS["qkv"], part)
                    ops.pool_qkv_post_partials(part, s, w["bqkv"], q, H, KV, HD, llm.rope_cos, llm.rope_sin, st, self.dk[li], self.dvt[li])
                    att = ops.attention_decode_batch(q, self.dk[li], self.dvt[li], st, bucket, H, KV, HD, scale)
                    s = ops.gemm_partials(att, w["wo"], self.SPLITS["o"], part)
                    ops.splitk_residual_rmsnorm(part, s, x, w["ln2"], eps, x, xn)
                    if self.SPLITS["gateup"] >= 2:
                        s = ops.gemm_partials(xn, w["wgu"], self.SPLITS["gateup"], part)
                        a = ops.splitk_swiglu(part, s, P, NGU)
                    elif self._tiled() is not None and self._tiled()[0][li] is not None:
                        a = ops.gemm_wtiled(xn, self._tiled()[0][li], act=ops.ACT_SWIGLU16)
                    else:
                        a = ops.gemm(xn, w["wgu"], act=ops.ACT_SWIGLU16)
                    s = ops.gemm_partials(a, w["wdown"], self.SPLITS["down"], part)
                    ops.splitk_residual_rmsnorm(part, s, x, llm.layers[li + 1]["ln1"] if li + 1 < len(llm.layers) else llm.norm, eps, x, xn)
                logits = ops.gemm_wtiled(xn, self._tiled()[1]) if (self._tiled() is not None and self._tiled()[1] is not None) else ops.gemm(xn, llm.lm_head)
            else:
                for li, w in enumerate(llm.layers):
                    qkv = ops.gemm(ops.rmsnorm(x, w["ln1"], c.rms_norm_eps), w["wqkv"], w["bqkv"])
                    ops.pool_qkv_post(qkv, H, KV, HD, llm.rope_cos, llm.rope_sin, st, self.dk[li], self.dvt[li])
                    att = ops.attention_decode_batch(qkv[:, :H * HD], self.dk[li], self.dvt[li], st, bucket, H, KV, HD, scale)
                    x = ops.gemm(att, w["wo"], residual=x)
                    a = ops.gemm(ops.rmsnorm(x, w["ln2"], c.rms_norm_eps), w["wgu"], act=ops.ACT_SWIGLU16)
                    x = ops.gemm(a, w["wdown"], residual=x)
                logits = ops.gemm(ops.rmsnorm(x, llm.norm, c.rms_norm_eps), llm.lm_head)
            ops.decode_argmax_accept(logits, None, st, self.plan, self.ids, self.stop, self.done, per_sequence_sets=True)
            return logits

    def step(self, use_graph: bool = True):
        """One token for every live sequence (all P rows are computed)."""
        bucket = self.kv_bucket()
        if not use_graph:
            out = self._step_device(bucket)
        else:
            # (the step's form is part of the key: FUSED_SPLITK / SPLITS may be set per pool, scripts/pool_bench.py and the A/B test do)
            key = (self.slot_rows, bucket, self.llm.rope_epoch, self.llm.rope_cos.data_ptr(), bool(self.FUSED_SPLITK), tuple(sorted(self.SPLITS.items())), bool(self.TILED_WEIGHTS))
            ent = self._graphs.get(key)
            if ent is None:
                with ops.graph_lock.capture(), torch.inference_mode(False):
                    snap = (self.state.clone(), self.plan.clone(), self.ids.clone(), self.done.clone())
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        self._step_device(bucket)      # warm-up (allocates scratch); rolled back below.  The K / V^T it appends land on
                    torch.cuda.current_stream().wait_stream(s)   # the rows the captured step rewrites with the same values
                    torch.cuda.synchronize()
                    for dst, src in zip((self.state, self.plan, self.ids, self.done), snap):
                        dst.copy_(src)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        logits = self._step_device(bucket)
                    ent = (g, logits)
                    self._graphs[key] = ent
            with ops.graph_lock.replay():
                ent[0].replay()
            out = ent[1]
        self.steps_run += 1
        for s in self.live:
            self.bound[s] += 1
        return out

    # ---- leave -----------------------------------------------------------------------------------------------------------------
    def snapshot(self, pinned: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """Asynchronous copy of (state, generated ids) to pinned host memory on the current stream -> (state, ids, event, live slots)."""
        cols = min(self.IDS_CAP, max((self.budget[s] for s in self.live), default=1))
        if pinned is None or pinned[1].numel() < self.P * cols:
            pinned = (torch.empty(self.P, 8, dtype=torch.int32).pin_memory(), torch.empty(self.P * max(cols, 64), dtype=torch.int32).pin_memory())
        pinned[0].copy_(self.state, non_blocking=True)
        host_ids = pinned[1][:self.P * cols].view(self.P, cols)            # contiguous on both sides: a plain asynchronous memcpy
        host_ids.copy_(self.ids[:, :cols].contiguous(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return pinned[0], host_ids, ev, dict(self.live), pinned

    def harvest(self, snap) -> List[Tuple[int, object, List[int]]]:
        """Sequences that had finished when `snap` was taken (and have not been collected yet): [(slot, tag, ids)]; their slots are free again."""
        st, ids, ev, live_then = snap[:4]
        ev.synchronize()
        fin, ngen = st[:, 3].tolist(), st[:, 4].tolist()          # one host read of the pinned snapshot, not two per live slot
        out = []
        for s, tag in live_then.items():
            if s in self.live and self.live[s] is tag and fin[s] == 1:
                out.append((s, tag, ids[s, :ngen[s]].tolist()))
                del self.live[s]
                self.free.append(s)
                self._set_users[self.slot_set[s]] -= 1
        return out

    def drain(self, use_graph: bool = True, poll: int = 8) -> List[Tuple[int, object, List[int]]]:
        """Step until every live sequence has stopped (synchronous convenience for tests / one-shot callers)."""
        out = []
        while self.live:
            for _ in range(poll):
                self.step(use_graph)
            out += self.harvest(self.snapshot())
        return out
