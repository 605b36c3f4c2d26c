"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement (plain torch, fp32 math on bf16-valued weights) of the Qwen2.5-VL language
model half of the hot path, following the reference's vendored file
  vlm_fo1/model/multimodal_encoder/qwen2_5_vl/modeling_qwen2_5_vl.py
    Qwen2RMSNorm                      :126-140
    Qwen2_5_VLRotaryEmbedding.forward :603-624
    apply_multimodal_rotary_pos_emb   :643-685
    repeat_kv / eager attention       :688-697, :738-802
    Qwen2MLP                          :627-640
    Qwen2_5_VLDecoderLayer            :1014-1095
    Qwen2_5_VLModel.forward           :1126-1242
    get_rope_index                    :1546-1701
and the splice of vlm_fo1/model/language_model/omchat_qwen2_5_vl.py:291-373.
Pinned in tests/test_oracle_llm.py against the reference's OWN vendored `Qwen2_5_VLModel` run in place on the CPU (prefill and the
KV-cache greedy decode path; oracle/reference_loader.py:vendored_llm builds it under the installed transformers 5 with two construction
shims — the rope initialiser key and pad_token_id — none in the arithmetic), and against the installed HF `Qwen2_5_VLTextModel`
(same architecture; the two produce bit-identical fp32 outputs, which is also how tests/golden/llm_ref.npz is generated).

State-dict keys are the checkpoint's (`layers.{i}.self_attn.q_proj.weight`, ...).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200      # vlm_fo1/constants.py:6
DEFAULT_REGION_INDEX = -300   # vlm_fo1/constants.py:19


def rmsnorm(x, w, eps):
    v = x.float()
    v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps)
    return w.float() * v


def rope_index(n_before: int, grid_hw_merged: Tuple[int, int], n_after: int):
    """Position ids [3, L] for  <text n_before> <image gh x gw merged tokens> <text n_after>
    (get_rope_index :1546-1701 for one image, t = 1; region tokens count as text)."""
    gh, gw = grid_hw_merged
    parts = [torch.arange(n_before).view(1, -1).expand(3, -1)]
    st = n_before
    t_idx = torch.zeros(gh * gw, dtype=torch.long)
    h_idx = torch.arange(gh).view(-1, 1).expand(-1, gw).flatten()
    w_idx = torch.arange(gw).view(1, -1).expand(gh, -1).flatten()
    parts.append(torch.stack([t_idx, h_idx, w_idx]) + st)
    nxt = int(parts[-1].max()) + 1
    parts.append(torch.arange(n_after).view(1, -1).expand(3, -1) + nxt)
    pos = torch.cat(parts, dim=1)
    delta = int(pos.max()) + 1 - pos.shape[1]
    return pos, delta


def mrope_cos_sin(pos: torch.Tensor, head_dim: int, theta: float, sections: Sequence[int]):
    """pos [3, L] -> section-selected cos, sin [L, head_dim] in fp32 (:609-618, :675-681)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = pos.float()[:, :, None] * inv[None, None, :]          # [3, L, hd/2]
    emb = torch.cat([fr, fr], dim=-1)                          # [3, L, hd]
    cos, sin = emb.cos(), emb.sin()
    sec = list(sections) * 2
    cs = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sn = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cs, sn


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def llm_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, pos: torch.Tensor, *, n_layers: int, n_heads: int,
                n_kv: int, head_dim: int, eps: float, theta: float, sections: Sequence[int], bf16_rope_tables: bool = True,
                return_all: bool = False):
    """x [L, d] embeddings -> final-norm hidden states [L, d] (fp32).  Causal, no cache."""
    L = x.shape[0]
    cos, sin = mrope_cos_sin(pos, head_dim, theta, sections)
    if bf16_rope_tables:  # reference :624 casts the tables to the activation dtype
        cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
    h = x.float()
    hs = []
    mask = torch.ones(L, L, dtype=torch.bool).tril()
    for i in range(n_layers):
        p = f"layers.{i}."
        r = rmsnorm(h, sd[p + "input_layernorm.weight"], eps)
        q = F.linear(r, sd[p + "self_attn.q_proj.weight"].float(), sd[p + "self_attn.q_proj.bias"].float())
        k = F.linear(r, sd[p + "self_attn.k_proj.weight"].float(), sd[p + "self_attn.k_proj.bias"].float())
        v = F.linear(r, sd[p + "self_attn.v_proj.weight"].float(), sd[p + "self_attn.v_proj.bias"].float())
        q = q.view(L, n_heads, head_dim)
        k = k.view(L, n_kv, head_dim)
        v = v.view(L, n_kv, head_dim)
        q = q * cos[:, None] + rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + rotate_half(k) * sin[:, None]
        rep = n_heads // n_kv
        kk = k.repeat_interleave(rep, dim=1)
        vv = v.repeat_interleave(rep, dim=1)
        att = torch.einsum("qhd,khd->hqk", q, kk) / math.sqrt(head_dim)
        att = att.masked_fill(~mask, float("-inf")).softmax(-1)
        o = torch.einsum("hqk,khd->qhd", att, vv).reshape(L, n_heads * head_dim)
        h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"].float())
        r = rmsnorm(h, sd[p + "post_attention_layernorm.weight"], eps)
        g = F.linear(r, sd[p + "mlp.gate_proj.weight"].float())
        u = F.linear(r, sd[p + "mlp.up_proj.weight"].float())
        h = h + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"].float())
        hs.append(h)
    out = rmsnorm(h, sd["norm.weight"], eps)
    return (out, hs) if return_all else out


def splice(input_ids: torch.Tensor, embed: torch.Tensor, image_tokens: torch.Tensor, region_tokens: Optional[torch.Tensor]):
    """input_ids [L] with sentinels -200 (expands to all image tokens) / -300 (one region token each)
    -> (embeds [L', d], n_before_image, n_after_image)   (omchat_qwen2_5_vl.py:317-373)."""
    rows = []
    ri = 0
    n_before = None
    for t in input_ids.tolist():
        if t == IMAGE_TOKEN_INDEX:
            n_before = len(rows)
            rows.extend(image_tokens.unbind(0))
        elif t == DEFAULT_REGION_INDEX:
            rows.append(region_tokens[ri])
            ri += 1
        else:
            rows.append(embed[t])
    out = torch.stack(rows)
    n_after = out.shape[0] - n_before - image_tokens.shape[0]
    return out, n_before, n_after


def random_llm_state(n_layers, d, n_heads, n_kv, head_dim, d_ff, vocab, seed=0, std=0.02):
    """Seeded bf16-valued random weights with the checkpoint's key names / shapes."""
    g = torch.Generator().manual_seed(seed)

    def w(*s, sc=std):
        return (torch.randn(*s, generator=g) * sc).bfloat16()

    sd = {"embed_tokens.weight": w(vocab, d), "norm.weight": (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()}
    for i in range(n_layers):
        p = f"layers.{i}."
        sd[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()
        sd[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16()
        sd[p + "self_attn.q_proj.weight"] = w(n_heads * head_dim, d)
        sd[p + "self_attn.q_proj.bias"] = w(n_heads * head_dim, sc=0.1)
        sd[p + "self_attn.k_proj.weight"] = w(n_kv * head_dim, d)
        sd[p + "self_attn.k_proj.bias"] = w(n_kv * head_dim, sc=0.1)
        sd[p + "self_attn.v_proj.weight"] = w(n_kv * head_dim, d)
        sd[p + "self_attn.v_proj.bias"] = w(n_kv * head_dim, sc=0.1)
        sd[p + "self_attn.o_proj.weight"] = w(d, n_heads * head_dim)
        sd[p + "mlp.gate_proj.weight"] = w(d_ff, d)
        sd[p + "mlp.up_proj.weight"] = w(d_ff, d)
        sd[p + "mlp.down_proj.weight"] = w(d, d_ff)
    return sd


def llm_forward_cached(sd: Dict[str, torch.Tensor], x: torch.Tensor, pos: torch.Tensor, cache: Optional[list] = None, *,
                       n_layers: int, n_heads: int, n_kv: int, head_dim: int, eps: float, theta: float, sections: Sequence[int],
                       bf16_rope_tables: bool = True):
    """Same arithmetic as llm_forward, but through a KV cache: x [Ln, d] are the NEW rows (a prompt on the first call, one
    token per greedy step afterwards), pos [3, Ln] their position ids.  `cache` is a list of per-layer (k, v) with RoPE
    already applied to k (what the reference's DynamicCache holds, modeling_qwen2_5_vl.py:770-776); pass the returned list
    back in.  Returns (final-norm hidden states of the new rows [Ln, d], cache).  Decode position = cache length + rope
    delta on all three axes (:1848-1860) is the caller's job (it builds `pos`)."""
    Ln = x.shape[0]
    cos, sin = mrope_cos_sin(pos, head_dim, theta, sections)
    if bf16_rope_tables:
        cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
    h = x.float()
    past = 0 if cache is None else cache[0][0].shape[0]
    new_cache = []
    qi = torch.arange(past, past + Ln)[:, None]
    ki = torch.arange(past + Ln)[None, :]
    mask = ki <= qi
    rep = n_heads // n_kv
    for i in range(n_layers):
        p = f"layers.{i}."
        r = rmsnorm(h, sd[p + "input_layernorm.weight"], eps)
        q = F.linear(r, sd[p + "self_attn.q_proj.weight"].float(), sd[p + "self_attn.q_proj.bias"].float()).view(Ln, n_heads, head_dim)
        k = F.linear(r, sd[p + "self_attn.k_proj.weight"].float(), sd[p + "self_attn.k_proj.bias"].float()).view(Ln, n_kv, head_dim)
        v = F.linear(r, sd[p + "self_attn.v_proj.weight"].float(), sd[p + "self_attn.v_proj.bias"].float()).view(Ln, n_kv, head_dim)
        q = q * cos[:, None] + rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + rotate_half(k) * sin[:, None]
        if cache is not None:
            k = torch.cat([cache[i][0], k], 0)
            v = torch.cat([cache[i][1], v], 0)
        new_cache.append((k, v))
        # GQA without materialising repeat_kv: group the query heads of one kv head
        qg = q.view(Ln, n_kv, rep, head_dim)
        att = torch.einsum("qgrd,kgd->grqk", qg, k) / math.sqrt(head_dim)
        att = att.masked_fill(~mask, float("-inf")).softmax(-1)
        o = torch.einsum("grqk,kgd->qgrd", att, v).reshape(Ln, n_heads * head_dim)
        h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"].float())
        r = rmsnorm(h, sd[p + "post_attention_layernorm.weight"], eps)
        g = F.linear(r, sd[p + "mlp.gate_proj.weight"].float())
        u = F.linear(r, sd[p + "mlp.up_proj.weight"].float())
        h = h + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"].float())
    return rmsnorm(h, sd["norm.weight"], eps), new_cache


def greedy_decode(sd: Dict[str, torch.Tensor], embeds: torch.Tensor, pos: torch.Tensor, rope_delta: int, n_new: int,
                  lm_head: Optional[torch.Tensor] = None, forced: Optional[Sequence[int]] = None, **kw):
    """Greedy generation through the cache (HF GenerationMixin greedy search + the reference's decode fast path,
    omchat_qwen2_5_vl.py:143-155).  Returns (ids [n_new], logits [n_new, V]).  With `forced` the i-th fed-back token is
    forced[i] instead of the argmax (teacher forcing on somebody else's choices); logits[i] are still the oracle's."""
    head = (lm_head if lm_head is not None else sd["embed_tokens.weight"]).float()
    hid, cache = llm_forward_cached(sd, embeds, pos, None, **kw)
    ids, logits = [], []
    last = hid[-1:]
    for i in range(n_new):
        lg = (last @ head.t())[0]
        logits.append(lg)
        t = int(lg.argmax())
        ids.append(t)
        if i + 1 == n_new:
            break
        fed = t if forced is None else int(forced[i])
        p = cache[0][0].shape[0] + rope_delta
        last, cache = llm_forward_cached(sd, sd["embed_tokens.weight"][fed:fed + 1].float(), torch.full((3, 1), p, dtype=torch.long), cache, **kw)
    return ids, torch.stack(logits)
