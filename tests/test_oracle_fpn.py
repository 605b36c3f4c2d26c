"""CPU test pinning oracle/fpn_oracle.py against the reference SimpleFP imported in place."""
import pytest
import torch

from oracle import fpn_oracle as FO
from oracle import hfre_oracle as HO


@pytest.mark.skipif(not HO.reference_available(), reason="/root/reference not present")
def test_fpn_oracle_matches_reference():
    _, SimpleFP, _ = HO.load_reference_hfre()
    dim, out = 128, 64
    m = SimpleFP(out_channels=out, norm="LN", square_pad=0, dim=dim, stride=14).eval()
    sd = FO.random_fpn_state(dim, out, seed=2)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    torch.manual_seed(0)
    x = torch.randn(1, dim, 10, 14).bfloat16().float()
    with torch.no_grad():
        ref = m(x)
    got = FO.fpn_forward(sd, x)
    assert [tuple(r.shape) for r in ref] == [(1, out, 40, 56), (1, out, 20, 28), (1, out, 10, 14), (1, out, 5, 7)]
    for g, r in zip(got, ref):
        torch.testing.assert_close(g, r, rtol=2e-4, atol=2e-4)


@pytest.mark.skipif(not HO.reference_available(), reason="/root/reference not present")
def test_fpn_true_key_shapes():
    _, SimpleFP, _ = HO.load_reference_hfre()
    m = SimpleFP(out_channels=512, norm="LN", square_pad=0, dim=1280, stride=14)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in FO.random_fpn_state().items()}
