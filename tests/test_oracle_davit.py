"""CPU test pinning oracle/davit_oracle.py against the reference DaViT imported in place."""
import types

import pytest
import torch

from oracle import davit_oracle as DO
from oracle import reference_loader as R

SMALL = dict(depths=(1, 1, 2, 1), dims=(64, 128, 256, 512), heads=(2, 4, 8, 16), groups=(2, 4, 8, 16),
             patch_size=(7, 3, 3, 3), patch_stride=(4, 2, 2, 2), patch_padding=(3, 1, 1, 1),
             patch_prenorm=(False, True, True, True), window=12)


@pytest.mark.skipif(not R.available(), reason="/root/reference not present")
@pytest.mark.parametrize("H,W", [(101, 125), (96, 96)])
def test_davit_oracle_matches_reference(H, W):
    davit, _ = R.vendored_davit()
    c = SMALL
    m = davit.DaViT(depths=c["depths"], embed_dims=c["dims"], num_heads=c["heads"], num_groups=c["groups"],
                    patch_size=c["patch_size"], patch_stride=c["patch_stride"], patch_padding=c["patch_padding"],
                    patch_prenorm=c["patch_prenorm"], window_size=c["window"], drop_path_rate=0.1).eval()
    sd = DO.random_davit_state(c, seed=H)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    torch.manual_seed(0)
    img = torch.randn(1, 3, H, W).bfloat16().float()
    with torch.no_grad():
        ref = m.forward_features(img)["image_features"]
    outs, sizes = DO.davit_forward(sd, img, c)
    for o, (h, w), r in zip(outs, sizes, ref):
        assert r.shape[2:] == (h, w)
        torch.testing.assert_close(o, r[0].permute(1, 2, 0).reshape(h * w, -1), rtol=3e-4, atol=3e-4)


def test_davit_large_key_shapes_match_reference_config():
    """The real configuration's parameter names/shapes (davit/configs.py:70-136)."""
    if not R.available():
        pytest.skip("/root/reference not present")
    davit, cfgs = R.vendored_davit()
    cfg = cfgs.model_configs["davit-large"]
    # names and shapes only: skip the (slow) random initialisation of 360 M parameters
    from unittest import mock
    noop = lambda t, *a, **k: t
    with mock.patch.object(davit, "trunc_normal_", noop), mock.patch("torch.nn.init.normal_", noop), \
            mock.patch("torch.nn.init.kaiming_uniform_", noop), mock.patch("torch.nn.init.uniform_", noop), \
            mock.patch("torch.nn.init.constant_", noop):
        m = davit.DaViT(depths=cfg["depths"], embed_dims=cfg["dim_embed"], num_heads=cfg["num_heads"], num_groups=cfg["num_groups"],
                        patch_size=cfg["patch_size"], patch_stride=cfg["patch_stride"], patch_padding=cfg["patch_padding"],
                        patch_prenorm=cfg["patch_prenorm"], window_size=cfg["window_size"])
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    with torch.device("meta"):
        sd = DO.random_davit_state(DO.DAVIT_LARGE)
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref_shapes
