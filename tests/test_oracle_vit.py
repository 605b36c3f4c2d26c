"""CPU test pinning oracle/vit_oracle.py against the reference's vendored Qwen2.5-VL ViT run in
place (sdpa attention) through the reference's own custom_forward + VisionFeaturesGather."""
import pytest
import torch

from oracle import reference_loader as R
from oracle import vit_oracle as VO


@pytest.mark.skipif(not R.available(), reason="/root/reference not present")
@pytest.mark.parametrize("gh,gw", [(6, 10), (16, 8), (18, 22)])
def test_vit_oracle_matches_reference(gh, gw):
    qwen = R.vendored_qwen()
    enc = R.vendored_vit_encoder()
    depth, d, heads, dff, out = 4, 160, 2, 96, 64   # head dim 80 like the real tower
    fullatt = [1, 3]
    cfg = qwen.Qwen2_5_VLVisionConfig(depth=depth, hidden_size=d, hidden_act="silu", intermediate_size=dff, num_heads=heads,
                                      in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2,
                                      window_size=112, out_hidden_size=out, fullatt_block_indexes=fullatt)
    cfg._attn_implementation = "sdpa"
    model = qwen.Qwen2_5_VisionTransformerPretrainedModel._from_config(cfg, attn_implementation="sdpa").eval().float()
    sd = VO.random_vit_state(depth, d, heads, dff, out, seed=gh * 100 + gw)
    missing = model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    S = gh * gw
    torch.manual_seed(1)
    pix = torch.randn(S, 1176).bfloat16().float()
    grid = torch.tensor([[1, gh, gw]])
    gather = enc.VisionFeaturesGather()
    model.vision_features_gather = gather
    with torch.no_grad():
        ref_tokens = enc.custom_forward(model, pix, grid)
        ref_maps = gather.extract_multi_level_features()[0]
    tokens, maps = VO.vit_forward(sd, pix, gh, gw, depth=depth, n_heads=heads, fullatt=fullatt)
    torch.testing.assert_close(tokens, ref_tokens, rtol=2e-4, atol=2e-4)
    assert len(maps) == len(ref_maps) == 2
    for m, r in zip(maps, ref_maps):
        assert r.shape == (1, d, gh, gw)
        torch.testing.assert_close(m, r[0].permute(1, 2, 0).reshape(S, d), rtol=2e-4, atol=2e-4)


def test_window_index_properties():
    for gh, gw in [(34, 46), (28, 36), (8, 8), (16, 16), (96, 96)]:
        widx, cu = VO.window_index(gh, gw)
        n = (gh // 2) * (gw // 2)
        assert sorted(widx.tolist()) == list(range(n)), "a permutation of the merge units"
        assert cu[0] == 0 and cu[-1] == gh * gw and (cu[1:] - cu[:-1]).max() <= 64 and (cu % 4 == 0).all()
