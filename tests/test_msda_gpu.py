"""Multi-scale deformable attention on the GPU (vlm_fo1_amd/csrc/msda.hip through the C-ABI and through the reference-shaped
MSDeformAttnFunction) against the CPU oracle (oracle/msda_ref.c) and the goldens made by the reference's own
ms_deform_attn_core_pytorch: the reference test's configuration and bars (ops/test.py: double: allclose at default tolerances;
float: rtol 1e-2 / atol 1e-3 — met with orders of magnitude to spare), UPN-shaped and ragged cases, the bf16 engine form, and a
full-size UPN geometry (900 queries, 800 x 1333 input pyramid) checked through linearity and a sampled oracle comparison."""
import os

import numpy as np
import pytest
import torch

import msda_cases as C

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "msda_ref.npz"))


def run(value, shapes, start, loc, w, via_function=False):
    from vlm_fo1_amd import ops
    sh = torch.as_tensor(shapes, dtype=torch.long).cuda()
    ls = torch.as_tensor(start, dtype=torch.long).cuda()
    if via_function:
        from detect_tools.upn.ops.functions import MSDeformAttnFunction
        return MSDeformAttnFunction.apply(value.cuda(), sh, ls, loc.cuda(), w.cuda(), 64).cpu()
    return ops.ms_deform_attn(value.cuda(), sh, ls, loc.cuda(), w.cuda()).cpu()


def test_reference_test_configuration():
    for tag, (value, shapes, start, loc, w) in C.reference_test_inputs().items():
        ref = torch.from_numpy(G[tag])
        for via in (False, True):
            got = run(value, shapes, start, loc, w, via)
            if value.dtype == torch.float64:
                assert torch.allclose(got, ref) and (got - ref).abs().max().item() <= 1e-16          # ops/test.py:42
            else:
                assert torch.allclose(got, ref, rtol=1e-2, atol=1e-3) and (got - ref).abs().max().item() <= 1e-8   # ops/test.py:57


def test_upn_shaped_and_ragged_cases_vs_golden_and_oracle():
    from oracle import msda_oracle as O
    for tag in C.CASES:
        value, shapes, start, loc, w = C.draw(tag)
        ref = torch.from_numpy(G[tag])
        got = run(value, shapes, start, loc, w)
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 2e-6, tag                     # fp32 kernel vs the fp64 reference evaluation
        orc = O.ms_deform_attn_forward(value, shapes, start, loc, w)
        assert (got - orc).abs().max().item() <= 1e-6, tag                     # same operation order; fma contraction may differ
        got64 = run(value.double(), shapes, start, loc.double(), w.double())
        assert (got64 - ref.double()).abs().max().item() <= 1e-7, tag


def test_bf16_value_engine_form():
    for tag in ("upn_decoder", "ragged"):
        value, shapes, start, loc, w = C.draw(tag)
        vb = value.bfloat16()
        got = run(vb, shapes, start, loc, w)
        assert got.dtype == torch.bfloat16
        from oracle import msda_oracle as O
        ref = O.ms_deform_attn_forward(vb.float(), shapes, start, loc, w)      # bf16-valued inputs, fp32 arithmetic, one rounding at the end
        err = (got.float() - ref).abs()
        assert (err <= 2.0 ** -8 * ref.abs() + 1e-6).all(), f"{tag}: max err {err.max():.4g}"


def test_full_size_upn_geometry_linearity_and_sampled_oracle():
    """800 x 1333 input -> Swin-L strides 8..128 (5 levels), 900 queries, 8 heads x 32, 4 points: too large for the scalar oracle in
    full; checked on 24 sampled queries against it, plus linearity in the value tensor and in the attention weights."""
    from oracle import msda_oracle as O
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
    start = [0]
    for h, ww in shapes[:-1]:
        start.append(start[-1] + h * ww)
    S = sum(h * w for h, w in shapes)
    N, M, D, Lq, L, P = 1, 8, 32, 900, 5, 4
    g = torch.Generator().manual_seed(77)
    value = torch.rand(N, S, M, D, generator=g) * 2 - 1
    value2 = torch.rand(N, S, M, D, generator=g) * 2 - 1
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1
    w = torch.rand(N, Lq, M, L, P, generator=g)
    w = w / w.sum((-1, -2), keepdim=True)
    a = run(value, shapes, start, loc, w)
    b = run(value2, shapes, start, loc, w)
    ab = run(value + 2 * value2, shapes, start, loc, w)
    assert (ab - (a + 2 * b)).abs().max().item() <= 2e-5
    assert (run(value, shapes, start, loc, 3 * w) - 3 * a).abs().max().item() <= 2e-5
    idx = torch.randperm(Lq, generator=g)[:24]
    ref = O.ms_deform_attn_forward(value, shapes, start, loc[:, idx].contiguous(), w[:, idx].contiguous())
    assert (a[:, idx] - ref).abs().max().item() <= 2e-6


def test_rejects_cpu_tensors_and_gradients():
    from detect_tools.upn.ops.functions import MSDeformAttnFunction
    from vlm_fo1_amd import ops
    value, shapes, start, loc, w = C.draw("ragged")
    sh, ls = torch.as_tensor(shapes, dtype=torch.long), torch.as_tensor(start, dtype=torch.long)
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn(value, sh, ls, loc, w)
    v = value.cuda().requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, sh.cuda(), ls.cuda(), loc.cuda(), w.cuda(), 64)
    with pytest.raises(NotImplementedError):
        out.sum().backward()
