"""CPU tests: pin the HFRE oracle (oracle/hfre_oracle.py, oracle/roi_align_ref.c) against
the committed golden vectors (made by the reference's own HFREModule) and, when
/root/reference is present, against the reference module run live; and check the
kernel's per-axis math (vlm_fo1_amd/csrc/hfre_math.h) through the host harness."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hfre_cases import CASES, checksum, make_case
from oracle import hfre_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FPN_STRIDES = [3.5, 7, 14, 28]


def oracle_out(case, roi_align=None):
    if case["fpn"]:
        return O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"],
                             region_dim=case["region_dim"], grid_hw=case["grid_hw"], vt_strides=FPN_STRIDES,
                             roi_align=roi_align)[0]
    return O.hfre_oracle(case["aux_maps"], case["boxes"], case["vt_maps"], case["vt_boxes"],
                         region_dim=case["region_dim"], grid_hw=case["grid_hw"], roi_align=roi_align)[0]


def load_golden(name):
    g = np.load(os.path.join(HERE, "golden", f"hfre_{name}.npz"))
    return g


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_golden(name):
    case = make_case(name)
    g = load_golden(name)
    assert str(g["checksum"]) == checksum(case), "seeded inputs drifted from the golden generator's"
    out = oracle_out(case)
    ref = torch.from_numpy(g["out"])
    got = out[:, torch.from_numpy(g["channels"]).long()]
    # same algorithm, same fp32 op order except torch's vectorised 49-element mean
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)


def test_roi_align_c_vs_torch_restatement():
    torch.manual_seed(3)
    x = torch.randn(1, 24, 37, 53)
    boxes = torch.tensor([[3., 4., 50., 60.], [0, 0, 212, 148], [10, 10, 10.5, 10.2], [100, 60, 130, 90],
                          [-5, -5, 20, 20], [200, 140, 230, 160], [0, 10, 0, 10]])
    a = O.roi_align_torch(x, boxes, 7, 0.25)
    c = O.roi_align_c(x, boxes, 7, 0.25)
    torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-6)
    a64 = O.roi_align_torch(x.double(), boxes.double(), 7, 0.25)
    assert (a64 - c.double()).abs().max() < 2e-5
    # strides: channels-last view must give the same numbers
    xc = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert torch.equal(O.roi_align_c(xc, boxes, 7, 0.25), c)
    # fused mean variant
    m = O.roi_align_c(x, boxes, 7, 0.25, mean=True)
    torch.testing.assert_close(m, c.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)


def test_roi_align_known_answers():
    """Hand-derivable values: constant map -> constant; linear ramp -> ROI centre value
    (bilinear taps reproduce linear functions exactly away from the border)."""
    x = torch.full((1, 2, 20, 30), 3.5)
    b = torch.tensor([[4., 4., 60., 40.]])
    assert torch.allclose(O.roi_align_c(x, b, 7, 0.25), torch.full((1, 2, 7, 7), 3.5))
    yy, xx = torch.meshgrid(torch.arange(20.), torch.arange(30.), indexing="ij")
    ramp = (2 * xx + 3 * yy).reshape(1, 1, 20, 30)
    r = O.roi_align_c(ramp, b, 7, 0.25, mean=True)
    # ROI in map coords: x in [1,15], y in [1,10]; mean of a linear function = value at the centre
    assert abs(r.item() - (2 * 8.0 + 3 * 5.5)) < 1e-4
    # sample outside [-1, W] contributes 0: box far outside -> exactly 0
    assert O.roi_align_c(x, torch.tensor([[500., 500., 600., 600.]]), 7, 0.25).abs().max() == 0


@pytest.mark.skipif(not O.reference_available(), reason="/root/reference not present")
@pytest.mark.parametrize("name", ["demo_nofpn", "edge_fpn"])
def test_oracle_matches_reference_live(name):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_golden import run_reference
    case = make_case(name)
    ref = run_reference(case)
    out = oracle_out(case)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6)
    # independent roi_align restatement (torch) through the same reference module path
    out_t = oracle_out(case, roi_align=O.roi_align_torch)
    torch.testing.assert_close(out_t, ref, rtol=1e-4, atol=1e-5)


# ---- kernel math through the host harness ------------------------------------------
class _Src(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("ld", ctypes.c_int32), ("roi_H", ctypes.c_int32), ("roi_W", ctypes.c_int32),
                ("scale", ctypes.c_float), ("box_space", ctypes.c_int32), ("out_offset", ctypes.c_int32)]


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "host_emul", "hfre_emul.cpp")
    so = os.path.join(HERE, "host_emul", "libhfre_emul.so")
    hdr = os.path.join(HERE, "..", "vlm_fo1_amd", "csrc", "hfre_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.hfre_emul_pool.restype = ctypes.c_long
    return lib


def emul_pool(lib, case, budget, dims_keep=64):
    keep = []
    srcs = []
    off = 0
    H0, W0 = case["aux_maps"][0].shape[2:]

    def add(m, roi_hw, scale, space):
        nonlocal off
        t = m[0, :dims_keep].permute(1, 2, 0).float().contiguous()  # [H,W,c] fp32, bf16-valued
        keep.append(t)
        srcs.append(_Src(t.data_ptr(), t.shape[0], t.shape[1], dims_keep, dims_keep, roi_hw[0], roi_hw[1], scale, space, off))
        off += dims_keep

    for m in case["aux_maps"]:
        add(m, (H0, W0), 0.25, 0)
    if case["fpn"]:
        for m, s in zip(case["fpn_maps"], FPN_STRIDES):
            add(m, m.shape[2:], 1.0 / s, 1)
    else:
        for m in case["vt_maps"]:
            add(m, m.shape[2:], 1 / 14, 1)
    arr = (_Src * len(srcs))(*srcs)
    boxes = case["boxes"].contiguous()
    N = boxes.shape[0]
    out = torch.zeros(N, off)
    sx, sy = case["vt_scale"]
    ns = lib.hfre_emul_pool(arr, len(srcs), ctypes.c_void_p(boxes.data_ptr()), N, ctypes.c_float(sx), ctypes.c_float(sy),
                            7, budget, ctypes.c_void_p(out.data_ptr()), off)
    assert ns >= 0, "slice count exceeded the grid bound"
    return out


@pytest.mark.parametrize("name", ["demo_fpn", "edge_fpn", "countbench30_fpn", "demo_nofpn"])
@pytest.mark.parametrize("budget", [1024, 48])
def test_kernel_math_vs_oracle(emul, name, budget):
    case = make_case(name)
    got = emul_pool(emul, case, budget)
    # oracle on the same first-64-channel slices, no position embedding
    sub = dict(case)
    sub["aux_maps"] = [m[:, :64] for m in case["aux_maps"]]
    key = "fpn_maps" if case["fpn"] else "vt_maps"
    sub[key] = [m[:, :64] for m in case[key]]
    kw = dict(region_dim=8 * 64, grid_hw=case["grid_hw"], apply_pos=False)
    if case["fpn"]:
        kw["vt_strides"] = FPN_STRIDES
    ref = O.hfre_oracle(sub["aux_maps"], case["boxes"], sub[key], case["vt_boxes"], **kw)[0]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


def test_axis_weights_sum_to_one(emul):
    """Every sample valid -> the per-axis weights of an ROI sum to 1 (partition of unity)."""
    w = (ctypes.c_float * 200)()
    for lo, hi in [(3.0, 150.5), (0.0, 799.0), (17.25, 17.5), (100.0, 100.0), (640.0, 800.0)]:
        n = emul.hfre_emul_axis(ctypes.c_float(lo), ctypes.c_float(hi), ctypes.c_float(0.25), 7, 200, w)
        assert n >= 1
        assert abs(sum(w) - 1.0) < 1e-5, (lo, hi, sum(w))


def test_oracle_variants_vs_reference_golden():
    """region LayerNorm / concat_aux_pos / vt-only: the oracle's restatement against the reference HFREModule's own outputs
    (tests/golden/hfre_variants.npz, made by tests/golden/make_hfre_variant_golden.py)."""
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_variant_golden import ln_params
    case = make_case("demo_fpn")
    g = np.load(os.path.join(HERE, "golden", "hfre_variants.npz"))
    assert str(g["checksum"]) == checksum(case)
    kw = dict(grid_hw=case["grid_hw"], vt_strides=[3.5, 7, 14, 28])
    out = O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"], region_dim=5888, region_ln=ln_params(), **kw)[0]
    torch.testing.assert_close(out, torch.from_numpy(g["ln"]), rtol=1e-5, atol=1e-5)
    out = O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"], region_dim=5888, pos_from="aux", **kw)[0]
    torch.testing.assert_close(out, torch.from_numpy(g["aux_pos"]), rtol=1e-5, atol=1e-5)
    out = O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"], region_dim=2048, vt_only=True, **kw)[0]
    torch.testing.assert_close(out, torch.from_numpy(g["vt_only"]), rtol=1e-5, atol=1e-5)
    for name, strategy in (("fm_pos", "feature_map_based"), ("hybrid", "hybrid")):      # 2-D table on the aux levels (:327-335)
        out = O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"], region_dim=5888, strategy=strategy, **kw)[0]
        torch.testing.assert_close(out, torch.from_numpy(g[name]), rtol=1e-5, atol=1e-5)


def test_vt_only_ignores_layernorm_and_strategy_like_the_reference():
    """the reference's vt-only branch (:293-317) returns before any region LayerNorm and adds the vt box embedding whenever
    apply_position_embedding is set, whatever the strategy: its outputs with LN on / with 'feature_map_based' equal plain vt-only."""
    import numpy as np
    g = np.load(os.path.join(HERE, "golden", "hfre_variants.npz"))
    assert np.array_equal(g["vt_only_ln"], g["vt_only"]) and np.array_equal(g["vt_only_fm_pos"], g["vt_only"])
    case = make_case("demo_fpn")
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_variant_golden import ln_params
    out = O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"], region_dim=2048, vt_only=True,
                        region_ln=ln_params(), grid_hw=case["grid_hw"], vt_strides=[3.5, 7, 14, 28])[0]
    torch.testing.assert_close(out, torch.from_numpy(g["vt_only_ln"]), rtol=1e-5, atol=1e-5)


def test_aux_only_reference_raises_and_extension_is_pinned_piecewise():
    """mm_use_vision_tower_region_feature=False (the reference's default VALUE): the reference's HFRE never binds `out_box_feat` and
    raises UnboundLocalError — recorded by the golden generator and, when /root/reference is present, re-run here.  The engine's
    aux-only extension = the reference's aux block (equal to the first 3840 channels of the reference's no-embedding output) + the
    reference's own sine embedding of the aux boxes (gen_sineembed_for_position, :55-103)."""
    import numpy as np
    g = np.load(os.path.join(HERE, "golden", "hfre_variants.npz"))
    assert str(g["aux_only_error"]) == "UnboundLocalError"
    case = make_case("demo_fpn")
    out = O.hfre_oracle(case["aux_maps"], case["boxes"], None, None, region_dim=3840, aux_only=True, grid_hw=case["grid_hw"])[0]
    nopos = O.hfre_oracle(case["aux_maps"], case["boxes"], None, None, region_dim=3840, aux_only=True, apply_pos=False, grid_hw=case["grid_hw"])[0]
    torch.testing.assert_close(nopos, torch.from_numpy(g["nopos"])[:, :3840], rtol=1e-5, atol=1e-5)
    H0, W0 = case["aux_maps"][0].shape[-2:]
    emb = O.box_pos_embed(case["boxes"].float(), W0 / 0.25, H0 / 0.25, 3840 // 4)
    torch.testing.assert_close(out, nopos + emb[0], rtol=0, atol=0)
    if O.reference_available():
        HFREModule, _, sine = O.load_reference_hfre()
        m = HFREModule(roi_output_size=7, region_feature_dim=3840, apply_position_embedding=True, use_vision_tower_region_feature=False,
                       aux_vision_tower_spatial_scale=0.25, aux_vision_tower_region_feature_dims=[256, 512, 1024, 2048])
        with pytest.raises(UnboundLocalError):
            m(case["aux_maps"], [case["boxes"].clone()])
        # the embedding half against the reference's own function
        b = case["boxes"].float().clone()
        b[:, [0, 2]] /= W0 / 0.25
        b[:, [1, 3]] /= H0 / 0.25
        b[:, 2] -= b[:, 0]; b[:, 3] -= b[:, 1]; b[:, 0] += b[:, 2] / 2; b[:, 1] += b[:, 3] / 2
        torch.testing.assert_close((out - nopos), sine(b.unsqueeze(0), 3840 // 4)[0], rtol=1e-5, atol=1e-6)


def test_roi_align_edge_vectors_hand_computed():
    """torchvision 0.21.0 `roi_align(aligned=False, sampling_ratio=-1)` edge cases with HAND-derived answers (the
    function is not vendored under /root/reference, so every HFRE golden runs through this restatement — an error here would be
    invisible to golden and oracle comparisons alike).  Map f(y, x) = 10 y + x: bilinear interpolation reproduces it exactly between
    pixel centres, so each expected value below is arithmetic on the published contract: roi = max(x2 - x1, 1); grid =
    ceil(roi / P); sample = x1 + (p + (i + .5) / grid) * roi / P; a sample with y < -1 or y > H contributes 0; y is clamped to >= 0;
    y_lo >= H - 1 collapses to the last row."""
    def ramp(H, W):
        yy, xx = torch.meshgrid(torch.arange(float(H)), torch.arange(float(W)), indexing="ij")
        return (10 * yy + xx).reshape(1, 1, H, W)

    f45 = ramp(4, 5)

    def one(box, x=f45, P=1):
        both = [O.roi_align_c(x, torch.tensor([box]), P, 1.0), O.roi_align_torch(x, torch.tensor([box]), P, 1.0)]
        torch.testing.assert_close(both[0], both[1], rtol=0, atol=1e-5)     # the two restatements agree on the edge cases too
        return both[0][0, 0]

    # A. box exactly on the map's right / bottom border (x2 = W = 5, y2 = H = 4): roi 2 x 2, grid 2 x 2, samples y in {2.5, 3.5},
    #    x in {3.5, 4.5}; y = 3.5 and x = 4.5 have lo >= size - 1 -> collapse to row 3 / column 4:
    #    f(2.5, 3.5) = 28.5, f(2.5, 4) = 29, f(3, 3.5) = 33.5, f(3, 4) = 34 -> mean 31.25
    assert abs(one([3., 2., 5., 4.]).item() - 31.25) < 1e-5
    # B. samples with y in (-1, 0) are VALID and clamp to row 0: box y in [-0.8, 0.2] -> roi_h = 1, one sample at y = -0.3 -> row 0;
    #    x samples 1.5, 2.5 -> mean 2.0.  y = -1.0 exactly is still valid (the test is y < -1); y = -2.1 is outside -> 0
    assert abs(one([1., -0.8, 3., 0.2]).item() - 2.0) < 1e-5
    assert abs(one([1., -1.5, 3., -0.5]).item() - 2.0) < 1e-5
    assert one([1., -2.6, 3., -1.6]).item() == 0.0
    # C. sub-pixel ROI (0.3 x 0.1 px): both extents are raised to 1 -> one sample at (y, x) = (1.3 + .5, 2.2 + .5) -> 18 + 2.7
    assert abs(one([2.2, 1.3, 2.5, 1.4]).item() - 20.7) < 1e-5
    # D. a sample exactly at y = H is valid (the test is y > H) and reads the last row; just beyond it contributes 0
    assert abs(one([0., 3.5, 1., 4.5]).item() - 30.5) < 1e-5         # y = 4.0 -> row 3, x = 0.5
    assert one([0., 3.6, 1., 4.6]).item() == 0.0                     # y = 4.1 > H
    # E. grid = 1 with the real output size 7: a 7 x 7 px ROI has 1 px bins, ONE sample per bin at the bin centre
    #    -> out[ph][pw] = 10 (ph + .5) + (pw + .5)
    f88 = ramp(8, 8)
    got = one([0., 0., 7., 7.], f88, P=7)
    ph, pw = torch.meshgrid(torch.arange(7.), torch.arange(7.), indexing="ij")
    torch.testing.assert_close(got, 10 * (ph + 0.5) + (pw + 0.5), rtol=0, atol=1e-5)
    #    7.5 px -> grid = ceil(7.5 / 7) = 2: samples at bin quarter points; a linear map -> still the bin-centre value while every
    #    sample stays below the last row / column (max sample = 7.5 * (6.75 / 7) = 7.232 > 7 = H - 1 collapses: excluded bins 6)
    got = one([0., 0., 7.5, 7.5], f88, P=7)
    bw = 7.5 / 7
    torch.testing.assert_close(got[:6, :6], (10 * (ph + 0.5) * bw + (pw + 0.5) * bw)[:6, :6], rtol=0, atol=1e-4)
    #    last bin: samples at 7.5 * 6.25 / 7 = 6.6964 (interpolates) and 7.2321 (lo = 7 >= H - 1 -> row 7 exactly)
    last = (7.5 * 6.25 / 7 + 7.0) / 2
    assert abs(got[6, 6].item() - (10 * last + last)) < 1e-4
    # F. the fused mean of the bins (what HFRE consumes, hybrid_finegrained_region_encoder.py:255,270,361) = mean of the 49 values
    m = O.roi_align_c(f88, torch.tensor([[0., 0., 7., 7.]]), 7, 1.0, mean=True)
    assert abs(m.item() - (10 * 3.5 + 3.5)) < 1e-4
