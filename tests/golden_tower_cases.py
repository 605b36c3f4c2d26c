"""Seeded cases shared by tests/golden/make_tower_goldens.py (reference side) and the golden tests (oracle / HIP side)."""
import torch

VIT = dict(depth=2, fullatt=(1,), grid=(10, 6), seed=10)
DAVIT = dict(hw=(96, 128), seed=1)
FPN = dict(grid=(6, 8), seed=4)
LLM = dict(layers=2, vocab=4096, L=48, seed=3, rope=(7, (3, 5), 48 - 7 - 15))


def vit_input():
    g = torch.Generator().manual_seed(3)
    gh, gw = VIT["grid"]
    return torch.randn(gh * gw, 1176, generator=g).bfloat16()


def davit_input():
    g = torch.Generator().manual_seed(96)
    H, W = DAVIT["hw"]
    return torch.randn(1, 3, H, W, generator=g).bfloat16()


def fpn_input():
    g = torch.Generator().manual_seed(5)
    gh, gw = FPN["grid"]
    return torch.randn(gh * gw, 1280, generator=g).bfloat16()


def llm_input():
    from oracle import llm_oracle as LO
    g = torch.Generator().manual_seed(0)
    x = torch.randn(LLM["L"], 2048, generator=g).bfloat16()
    pos, _ = LO.rope_index(*LLM["rope"])
    return x, pos
