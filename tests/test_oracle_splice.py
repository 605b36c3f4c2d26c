"""The splice pinned to the reference's OWN function: tests/golden/splice_ref.npz holds what
`OmChatQwen25VLForCausalLM.prepare_inputs_labels_for_qwen2_5_vl_multimodal` (omchat_qwen2_5_vl.py:135-463, run in place by
tests/golden/make_splice_golden.py) returns on the seeded prompts of tests/splice_cases.py — one prompt, a 100-region prompt, and a
ragged batch of three (the reference right-pads to the longest; the engine packs without padding).  Checked here, on the CPU:
the oracle's splice + rope index, and the engine's host planner (QwenLLM.plan_inputs / plan_batch: gather plan, position ids,
rope delta, packed layout)."""
import os
import types

import numpy as np
import torch

import splice_cases as C
from oracle import llm_oracle as LO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splice_ref.npz"))


def _ref(name):
    return {k: torch.from_numpy(G[f"{name}.{k}"]) for k in ("embeds", "position_ids", "attention_mask", "rope_deltas", "cache_position")}


def _engine_llm():
    from vlm_fo1_amd import llm as LL
    cfg = types.SimpleNamespace(vocab_size=C.VOCAB, head_dim=128, rope_theta=1e6, mrope_section=(16, 24, 24))
    fake = types.SimpleNamespace(cfg=cfg, PACK_ALIGN=LL.QwenLLM.PACK_ALIGN)
    fake.plan_inputs = types.MethodType(LL.QwenLLM.plan_inputs, fake)
    fake.plan_batch = types.MethodType(LL.QwenLLM.plan_batch, fake)
    return fake


def _gather(plan, table, img, reg):
    rows = []
    for kind, idx in plan.tolist():
        rows.append(table[idx] if kind == 0 else (img[idx] if kind == 1 else reg[idx]))
    return torch.stack(rows)


def test_oracle_and_engine_planner_equal_the_reference_splice():
    table = C.embed_table()
    eng = _engine_llm()
    for name, batch in C.cases().items():
        ref = _ref(name)
        B, Lmax = ref["embeds"].shape[:2]
        assert B == len(batch) and torch.equal(ref["cache_position"], torch.arange(Lmax))
        for b, p in enumerate(batch):
            ids = torch.tensor(p["ids"])
            n_img = p["grid_merged"][0] * p["grid_merged"][1]
            L = len(p["ids"]) - 1 + n_img                                # <image> expands, every <region> is one row
            # --- the reference's padding: real rows first, zeros after, mask accordingly
            assert ref["attention_mask"][b, :L].all() and not ref["attention_mask"][b, L:].any()
            assert not ref["embeds"][b, L:].any()
            # --- oracle
            emb, nb, na = LO.splice(ids, table, p["image_tokens"], p["region_tokens"])
            assert emb.shape[0] == L and torch.equal(emb, ref["embeds"][b, :L])
            pos, delta = LO.rope_index(nb, p["grid_merged"], na)
            assert torch.equal(pos, ref["position_ids"][:, b, :L])
            # the reference's delta is relative to the PADDED length (decode position = cache_position + delta, cache positions count
            # the padding); the engine keeps no padding: same next position
            assert delta == int(ref["rope_deltas"][b]) + (Lmax - L)
            # --- engine host planner, one prompt
            plan, pos_e, delta_e = eng.plan_inputs(p["ids"], n_img, p["n_regions"], p["grid_merged"])
            assert torch.equal(_gather(plan, table, p["image_tokens"], p["region_tokens"]), ref["embeds"][b, :L])
            assert torch.equal(pos_e, ref["position_ids"][:, b, :L]) and delta_e == delta
        # --- engine host planner, the whole batch packed (no padding to the longest; PACK_ALIGN dummy rows after each sequence)
        hp = eng.plan_batch([p["ids"] for p in batch], [p["grid_merged"][0] * p["grid_merged"][1] for p in batch],
                            [p["n_regions"] for p in batch], [p["grid_merged"] for p in batch])
        img_all = torch.cat([p["image_tokens"] for p in batch])
        reg_all = torch.cat([p["region_tokens"][:p["n_regions"]] for p in batch]) if any(p["n_regions"] for p in batch) else torch.zeros(1, C.D)
        packed = _gather(hp["plan"], table, img_all, reg_all)
        for b, (off, L, Lp) in enumerate(hp["seqs"]):
            assert torch.equal(packed[off:off + L], ref["embeds"][b, :L])
            assert torch.equal(hp["pos"][b], ref["position_ids"][:, b, :L])
            assert hp["delta"][b] == int(ref["rope_deltas"][b]) + (Lmax - L)
            assert int(hp["last"][b, 1]) == off + L - 1
