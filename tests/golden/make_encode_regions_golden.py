"""Generates tests/golden/encode_regions_ref.npz: the reference's OWN `OmChatQwen25VLForCausalLM.encode_regions`
(vlm_fo1/model/language_model/omchat_qwen2_5_vl.py:75-128: dummy box, fp32 boxes, vt-space box scaling, HFREModule call, cast to the
tower dtype, mm_projector_aux) run in place on the CPU as an unbound method on a stand-in `self`, on seeded cases of tests/hfre_cases.py.
`object_vp_extractor` is the reference's own HFREModule; `torchvision.ops.roi_align` (not installed, not vendored) is the restatement in
oracle/roi_align_ref.c; the aux tower and the ViT maps are the seeded tensors; `mm_projector_aux` is a seeded Linear (its arithmetic is
not the point: the glue around it is).

    python tests/golden/make_encode_regions_golden.py
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") not in (ROOT, HERE)]
sys.path.insert(0, "/root/reference")                    # `vlm_fo1` = the reference's package
sys.path.insert(1, ROOT)                                 # `oracle` (this repo's drop-in `vlm_fo1` is shadowed by the line above)
sys.path.insert(2, os.path.join(ROOT, "tests"))

import numpy as np          # noqa: E402
import torch                # noqa: E402
import transformers         # noqa: E402,F401

from oracle.hfre_oracle import roi_align_c              # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _nope(*a, **k):
    raise RuntimeError("stub: not on the path under test")


tv = stub("torchvision"); tv.__path__ = []
tv.ops = stub("torchvision.ops", roi_align=roi_align_c)
tv.transforms = stub("torchvision.transforms", ToPILImage=object, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
stub("torchvision.transforms.functional")
stub("transformers.models.qwen2_vl.image_processing_qwen2_vl", Qwen2VLImageProcessor=object)
timm = stub("timm"); timm.__path__ = []
tm = stub("timm.models"); tm.__path__ = []
tm.layers = stub("timm.models.layers", DropPath=torch.nn.Identity, trunc_normal_=_nope)
timm.layers = stub("timm.layers", LayerNorm=torch.nn.LayerNorm, LayerNorm2d=torch.nn.LayerNorm, DropPath=torch.nn.Identity, trunc_normal_=_nope)
stub("timm.models.regnet", RegStage=object)
stub("timm.models.resnet", Bottleneck=object)

import vlm_fo1.model.language_model.omchat_qwen2_5_vl as O                                   # noqa: E402
from vlm_fo1.model.multimodal_visual_prompt_encoder.hybrid_finegrained_region_encoder import HFREModule   # noqa: E402
import encode_regions_cases as EC                                                             # noqa: E402

assert O.__file__.startswith("/root/reference/")


def run(case):
    fpn = case["fpn"]
    torch.manual_seed(0)
    hfre = HFREModule(roi_output_size=7, region_feature_dim=case["region_dim"], apply_position_embedding=True,
                      pos_embedding_strategy="bbox_based", use_vt_region_feature_only=False, use_vision_tower_region_feature=True,
                      region_feature_combination="concat", apply_region_layer_norm=False,
                      vision_tower_region_feature_dim=2048 if fpn else 5120, vision_tower_spatial_scale=1 / 14,
                      use_simpleFPN_for_vt=fpn, aux_vision_tower_spatial_scale=0.25, aux_vision_tower_region_feature_dims=[256, 512, 1024, 2048])
    gh, gw = case["grid_hw"]
    if fpn:
        class _Fixed(torch.nn.Module):
            def forward(self, x):
                return case["fpn_maps"]
        hfre.simple_fpn = _Fixed()
        vt_levels = [torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16)] * 4     # with the FPN only the LAST capture is handed on (:82-85)
    else:
        vt_levels = case["vt_maps"]
    proj = torch.nn.Linear(case["region_dim"], EC.D_OUT).to(torch.bfloat16)
    proj.weight.data.copy_(EC.projector(case["region_dim"])[0]); proj.bias.data.copy_(EC.projector(case["region_dim"])[1])
    aux_out = [{"image_features": case["aux_maps"], "last_feat": case["aux_maps"][-1]}]
    model = types.SimpleNamespace(get_vision_tower_aux=lambda: (lambda images: aux_out), object_vp_extractor=hfre, mm_projector_aux=proj)
    cfg = types.SimpleNamespace(mm_use_vision_tower_region_feature=True, mm_use_simpleFPN_for_vt=fpn)
    fake = types.SimpleNamespace(config=cfg, get_model=lambda: model)
    H, W = case["img"]
    images = [torch.zeros(1, 3, H, W)]                                          # aux tensors: only their spatial size is read (:95)
    vt_size = [torch.tensor([gh * 14, gw * 14])]                                # (h, w) of the primary tower's input (:96, from grid_thw x patch)
    with torch.no_grad():
        out = O.OmChatQwen25VLForCausalLM.encode_regions(fake, images, [case["boxes_in"]], [vt_levels], vt_size)
    assert len(out) == 1
    return out[0]


if __name__ == "__main__":
    res = {}
    for name in EC.NAMES:
        case = EC.make(name)
        out = run(case)
        res[name] = out.float().numpy()
        print(name, tuple(out.shape), out.dtype)
    np.savez_compressed(os.path.join(HERE, "encode_regions_ref.npz"), **res)
    print("wrote tests/golden/encode_regions_ref.npz")
