"""Shared by scripts/hfre_sweep.py and scripts/hfre_ab.py: HFRE inputs on the 640x480 / 100-box geometry for B images per call."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_hfre_gpu import _full_size_case, to_dev            # noqa: E402
from vlm_fo1_amd.hfre import HFREModule                      # noqa: E402


def build(B, n_boxes=100):
    devs = [to_dev(_full_size_case(480, 640, n_boxes, 100 + i)) for i in range(B)]
    def stack(key, lvl):
        tm = torch.stack([d[key][lvl].permute(0, 2, 3, 1)[0] for d in devs]).contiguous()
        return tm[:1].permute(0, 3, 1, 2)
    aux = [stack("aux_maps", i) for i in range(4)]
    fpn = [stack("fpn_maps", i) for i in range(4)]
    boxes = torch.cat([d["boxes"] for d in devs]).contiguous()
    bi = torch.cat([torch.full((d["boxes"].shape[0],), i, dtype=torch.int32) for i, d in enumerate(devs)]).cuda()
    gh, gw = devs[0]["grid_hw"]
    m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, use_vision_tower_region_feature=True,
                   vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True, simple_fpn=lambda x: fpn)
    vt_in = torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(boxes.shape[0], 5888, dtype=torch.float32, device="cuda")
    def call():
        if B == 1:
            m(aux, [boxes], vt_in, None, vt_scale=devs[0]["vt_scale"], out=out)
        else:
            m(aux, [boxes], vt_in, None, vt_scale=devs[0]["vt_scale"], out=out, batch=B, box_image=bi)
    return m, call, out
