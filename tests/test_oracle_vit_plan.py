"""The ViT's HOST index plan (vlm_fo1_amd/vit.py GridPlan: window permutation, window segments, 2-D rope angles, raster gather plan)
pinned to the reference's own index functions run in place on the CPU — `get_window_index` / `rot_pos_emb`
(modeling_qwen2_5_vl.py:465-504, 436-463), the rotary reordering of `forward` (:528-535) and the capture un-windowing of
`VisionFeaturesGather.extract_multi_level_features` (qwen2_5_vl_encoder.py:37-80).  Pure integer / fp32 table work: exact."""
import types

import pytest
import torch

from oracle import reference_loader as R

pytestmark = pytest.mark.skipif(not R.available(), reason="/root/reference not present")

GRIDS = [(34, 46), (28, 36), (8, 8), (6, 10), (96, 96), (30, 52)]


def _ref_plan(gh, gw):
    ref = R.vendored_qwen()
    VT = ref.Qwen2_5_VisionTransformerPretrainedModel
    fake = types.SimpleNamespace(window_size=112, spatial_merge_size=2, patch_size=14, spatial_merge_unit=4,
                                 rotary_pos_emb=ref.Qwen2_5_VisionRotaryEmbedding(80 // 2))
    thw = torch.tensor([[1, gh, gw]])
    window_index, cu_window = VT.get_window_index(fake, thw)
    rot = VT.rot_pos_emb(fake, thw)                                                     # [S, 40] in merge-block order
    S = gh * gw
    rot = rot.reshape(S // 4, 4, -1)[window_index].reshape(S, -1)                         # forward :528-531
    cu = torch.unique_consecutive(torch.tensor(cu_window, dtype=torch.int32)).tolist()    # forward :520-526
    return window_index, cu, rot


@pytest.mark.parametrize("gh,gw", GRIDS)
def test_grid_plan_equals_reference_index_functions(gh, gw):
    from vlm_fo1_amd.vit import GridPlan, ViTConfig
    g = GridPlan(gh, gw, ViTConfig(), "cpu")
    window_index, cu, rot = _ref_plan(gh, gw)
    S = gh * gw
    # window permutation: window-order row r reads merge-block-order row perm[r] (forward :514-518)
    perm_ref = (window_index[:, None] * 4 + torch.arange(4)[None, :]).reshape(-1)
    assert torch.equal(g.plan_in[:, 1].long(), perm_ref)
    assert list(g.cu_window) == cu
    assert torch.equal(g.cos, rot.cos()) and torch.equal(g.sin, rot.sin())
    # raster plan: run the reference's capture un-windowing on a map whose "feature" is the window-order row number
    enc = R.vendored_vit_encoder()
    gather = enc.VisionFeaturesGather()
    gather.grid_thw = torch.tensor([[1, gh, gw]])
    gather.window_index = window_index
    gather.merge_size = 2
    gather.features_list = [torch.arange(S, dtype=torch.float32)[:, None]]
    m = gather.extract_multi_level_features()[0][0]                                       # [1, 1, gh, gw]
    assert torch.equal(g.plan_raster[:, 1].long(), m[0, 0].reshape(-1).long())
    # image tokens leave the merger in window order and are put back in merge-unit order by argsort(window_index) (:553-556)
    assert torch.equal(g.plan_tokens[:, 1].long(), torch.argsort(window_index))
