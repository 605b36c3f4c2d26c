"""The C restatement of multi-scale deformable attention (oracle/msda_ref.c) against goldens made by the reference's own
ms_deform_attn_core_pytorch (tests/golden/make_msda_golden.py): the reference test's configuration (ops/test.py) in fp64 and fp32,
UPN-shaped cases with out-of-range sampling locations, and a ragged case.  The bars are the reference test's own
(ops/test.py:42,57: allclose at default tolerances in double, rtol 1e-2 / atol 1e-3 in float) — met with orders of magnitude to spare."""
import os

import numpy as np
import torch

import msda_cases as C
from oracle import msda_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "msda_ref.npz"))


def test_reference_test_configuration():
    for tag, (value, shapes, start, loc, w) in C.reference_test_inputs().items():
        got = O.ms_deform_attn_forward(value, shapes, start, loc, w)
        ref = torch.from_numpy(G[tag])
        assert got.shape == ref.shape == (1, 2, 4) and got.dtype == ref.dtype
        if value.dtype == torch.float64:
            assert torch.allclose(got, ref), (got - ref).abs().max()                 # the reference's bar (ops/test.py:42)
            assert (got - ref).abs().max().item() <= 1e-17
        else:
            assert torch.allclose(got, ref, rtol=1e-2, atol=1e-3)                    # the reference's bar (ops/test.py:57)
            assert (got - ref).abs().max().item() <= 1e-9


def test_upn_shaped_and_ragged_cases():
    for tag in C.CASES:
        value, shapes, start, loc, w = C.draw(tag)
        ref = torch.from_numpy(G[tag]).double()
        got64 = O.ms_deform_attn_forward(value.double(), shapes, start, loc.double(), w.double())
        assert got64.shape == ref.shape
        assert (got64 - ref).abs().max().item() <= 1e-7, tag                          # golden = fp64 result rounded to fp32
        got32 = O.ms_deform_attn_forward(value, shapes, start, loc, w)
        assert (got32.double() - ref).abs().max().item() <= 2e-6, tag


def test_out_of_range_samples_contribute_nothing():
    """Samples with h_im <= -1 or >= H (w alike) are skipped; samples in (-1, 0) x ... see only the in-range taps
    (ms_deform_im2col_cuda.cuh:279, :50-73)."""
    value = torch.ones(1, 6, 1, 1)
    shapes, start = [(2, 3)], [0]
    loc = torch.tensor([[-0.4, 0.5], [0.5, 1.6], [0.5, 0.5], [0.0, 0.0]]).view(1, 4, 1, 1, 1, 2)   # (x, y)
    w = torch.ones(1, 4, 1, 1, 1)
    got = O.ms_deform_attn_forward(value, shapes, start, loc, w).flatten()
    # x = -0.4 -> w_im = -1.7: skipped;  y = 1.6 -> h_im = 2.7 >= H: skipped;  centre: 1;  corner (0, 0): h_im = w_im = -0.5 -> quarter weight
    assert torch.allclose(got, torch.tensor([0.0, 0.0, 1.0, 0.25]))
