"""ops.zero_framed / ops.conv3x3_padded (round 6): persistent zero-framed scratch instead of a zero fill per pass — host logic only (CPU tensors):
one buffer per key, bounded count of persistent keys, ragged plans keep the per-call fill."""
import torch

from vlm_fo1_amd import ops


def _cpu_owner(monkeypatch):
    """the scratch pool keys its buffers by engine owner (or by HIP stream without one): give the CPU test an owner"""
    monkeypatch.setattr(ops._ws_tls, "owner", object(), raising=False)


def test_zero_framed_buffers_are_per_key_and_bounded(monkeypatch):
    _cpu_owner(monkeypatch)
    monkeypatch.setattr(ops, "_zero_framed_kinds", {})
    monkeypatch.setattr(ops, "ZERO_FRAMED_MAX", 3)
    a = ops.zero_framed(("t", 1), 4, 8, "cpu")
    assert a.shape == (4, 8) and a.dtype == torch.bfloat16 and not a.any()
    a[1:3, 2:6] = 1                                     # "interior" writes of a pass
    b = ops.zero_framed(("t", 1), 4, 8, "cpu")
    assert b.data_ptr() == a.data_ptr() and float(b[0].abs().sum()) == 0 and float(b[1, 2]) == 1       # same storage, frame still zero
    c = ops.zero_framed(("t", 2), 4, 8, "cpu")
    assert c.data_ptr() != a.data_ptr() and not c.any()
    ops.zero_framed(("t", 3), 4, 8, "cpu")
    d1, d2 = ops.zero_framed(("t", 4), 4, 8, "cpu"), ops.zero_framed(("t", 4), 4, 8, "cpu")              # past the bound: fresh fills
    assert d1.data_ptr() != d2.data_ptr() and not d1.any() and not d2.any()
    monkeypatch.setenv("FO1_ZERO_FRAMED", "0")
    e = ops.zero_framed(("t", 1), 4, 8, "cpu")
    assert e.data_ptr() != a.data_ptr() and not e.any()


def test_conv_padded_map_is_persistent_for_uniform_plans_only(monkeypatch):
    _cpu_owner(monkeypatch)
    monkeypatch.setattr(ops, "_zero_framed_kinds", {})
    uni = ops.Conv3x3Plan(((6, 5),) * 3, 1, 64, "cpu")
    rag = ops.Conv3x3Plan(((6, 5), (4, 7)), 1, 64, "cpu")
    assert uni.uniform and not rag.uniform and uni.serial != rag.serial
    p1, p2 = ops.conv3x3_padded(uni, "cpu"), ops.conv3x3_padded(uni, "cpu")
    assert p1.shape == (uni.pad_rows, 64) and p1.data_ptr() == p2.data_ptr()
    r1, r2 = ops.conv3x3_padded(rag, "cpu"), ops.conv3x3_padded(rag, "cpu")
    assert r1.shape == (rag.pad_rows, 64) and r1.data_ptr() != r2.data_ptr()
    # the rows a pass writes are the plan's rowmap; everything else is the frame
    p1[uni.rowmap.long()] = 1
    frame = torch.ones(uni.pad_rows, dtype=torch.bool)
    frame[uni.rowmap.long()] = False
    assert not ops.conv3x3_padded(uni, "cpu")[frame].any()
