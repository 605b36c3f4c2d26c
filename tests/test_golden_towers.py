"""CPU: the oracles at TRUE channel widths against outputs of the REFERENCE's own modules committed as goldens
(tests/golden/{vit,davit,fpn,llm}_ref.npz, made by tests/golden/make_tower_goldens.py from /root/reference + HF).  Needs neither
/root/reference nor a GPU, so it also runs on the GPU box; weights come from the CPU-seeded random_*_state helpers."""
import os

import numpy as np
import torch

from golden_tower_cases import DAVIT, FPN, LLM, VIT, davit_input, fpn_input, llm_input, vit_input
from oracle import davit_oracle as DO, fpn_oracle as FO, llm_oracle as LO, vit_oracle as VO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name)).items()}


def test_vit_oracle_vs_reference_golden():
    c, ref = VIT, gold("vit_ref.npz")
    sd = VO.random_vit_state(c["depth"], 1280, 16, 3420, 2048, seed=c["seed"])
    gh, gw = c["grid"]
    tokens, maps = VO.vit_forward(sd, vit_input().float(), gh, gw, depth=c["depth"], n_heads=16, fullatt=c["fullatt"])
    torch.testing.assert_close(tokens, ref["tokens"], rtol=3e-4, atol=3e-4)
    torch.testing.assert_close(maps[-1], ref["last_map"], rtol=3e-4, atol=3e-4)


def test_davit_oracle_vs_reference_golden():
    ref = gold("davit_ref.npz")
    sd = DO.random_davit_state(DO.DAVIT_LARGE, seed=DAVIT["seed"])
    outs, sizes = DO.davit_forward(sd, davit_input().float())
    assert [list(s) for s in sizes] == ref["sizes"].tolist()
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, ref[f"stage{i}"], rtol=5e-4, atol=5e-4)


def test_fpn_oracle_vs_reference_golden():
    ref = gold("fpn_ref.npz")
    sd = FO.random_fpn_state(seed=FPN["seed"])
    gh, gw = FPN["grid"]
    outs = FO.fpn_forward(sd, fpn_input().float().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
    for i, o in enumerate(outs):
        torch.testing.assert_close(o[0].permute(1, 2, 0).reshape(-1, 512), ref[f"level{i}"], rtol=3e-4, atol=3e-4)


def test_llm_oracle_vs_hf_golden():
    c, ref = LLM, gold("llm_ref.npz")
    sd = LO.random_llm_state(c["layers"], 2048, 16, 2, 128, 11008, c["vocab"], seed=c["seed"])
    x, pos = llm_input()
    got = LO.llm_forward(sd, x.float(), pos, bf16_rope_tables=False, n_layers=c["layers"], n_heads=16, n_kv=2, head_dim=128, eps=1e-6,
                         theta=1e6, sections=(16, 24, 24))
    torch.testing.assert_close(got, ref["hidden"], rtol=3e-4, atol=3e-4)
