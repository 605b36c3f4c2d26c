"""Seeded cases of the encode_regions parity test (a6): tests/hfre_cases.py maps + the inputs encode_regions itself consumes (caller
boxes in aux space, possibly none; image sizes).  Shared by tests/golden/make_encode_regions_golden.py and tests/test_oracle_encode_regions.py."""
import torch

from hfre_cases import CASES, make_case

D_OUT = 48
NAMES = ["demo_fpn", "demo_nofpn", "edge_fpn", "demo_fpn_no_boxes"]


def projector(region_dim):
    g = torch.Generator().manual_seed(region_dim)
    return (torch.randn(D_OUT, region_dim, generator=g) * 0.02).bfloat16(), (torch.randn(D_OUT, generator=g) * 0.1).bfloat16()


def make(name):
    base = name.replace("_no_boxes", "")
    c = make_case(base)
    c["img"] = CASES[base]["img"]
    c["boxes_in"] = None if name.endswith("_no_boxes") else c["boxes"].to(torch.float64)     # (the reference casts to fp32 itself, :93)
    return c
