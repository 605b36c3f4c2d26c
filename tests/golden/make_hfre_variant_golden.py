"""Generates tests/golden/hfre_variants.npz: the REFERENCE's own HFREModule (imported in place from /root/reference; roi_align = the
restatement in oracle/roi_align_ref.c, as in make_hfre_golden.py) on the seeded 'demo_fpn' case, in the configurations the engine
supports beyond the default one:
  ln          apply_region_layer_norm=True      (aux_region_norm / vt_region_norm with seeded weights; reference :365-372)
  aux_pos     region_feature_combination='concat_aux_pos'   (box embedding from the aux boxes; :443-455)
  vt_only     use_vt_region_feature_only=True   (:293-317; region_feature_dim = 2048)
  fm_pos      pos_embedding_strategy='feature_map_based'    (2-D sine table added to every aux level, no box embedding; :327-335)
  hybrid      pos_embedding_strategy='hybrid'               (both)
  vt_only_ln      vt-only + apply_region_layer_norm=True    (the vt-only branch returns BEFORE any LayerNorm, :293-317: == vt_only)
  vt_only_fm_pos  vt-only + 'feature_map_based'             (the vt-only branch tests apply_position_embedding alone: == vt_only)
  nopos       apply_position_embedding=False                (the pure [aux | vt] pooled blocks; the aux block pins the aux-only extension)
  aux_only_error  use_vision_tower_region_feature=False     (the reference's default value): the exception the reference raises —
                  `out_box_feat` is never bound on that path, UnboundLocalError for every input

    python tests/golden/make_hfre_variant_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hfre_cases import checksum, make_case  # noqa: E402
from oracle.hfre_oracle import load_reference_hfre  # noqa: E402


def ln_params(seed=123):
    g = torch.Generator().manual_seed(seed)
    return dict(aux_w=1 + 0.2 * torch.randn(3840, generator=g), aux_b=0.1 * torch.randn(3840, generator=g),
                vt_w=1 + 0.2 * torch.randn(2048, generator=g), vt_b=0.1 * torch.randn(2048, generator=g))


def run(case, variant):
    HFREModule, SimpleFP, _ = load_reference_hfre()
    kw = dict(roi_output_size=7, region_feature_dim=case["region_dim"], apply_position_embedding=True, pos_embedding_strategy="bbox_based",
              use_vt_region_feature_only=False, use_vision_tower_region_feature=True, region_feature_combination="concat",
              apply_region_layer_norm=False, vision_tower_region_feature_dim=2048, vision_tower_spatial_scale=1 / 14,
              use_simpleFPN_for_vt=True, aux_vision_tower_spatial_scale=0.25, aux_vision_tower_region_feature_dims=[256, 512, 1024, 2048])
    if variant == "ln":
        kw["apply_region_layer_norm"] = True
    elif variant == "aux_pos":
        kw["region_feature_combination"] = "concat_aux_pos"
    elif variant == "vt_only":
        kw["use_vt_region_feature_only"] = True
        kw["region_feature_dim"] = 2048
    elif variant == "fm_pos":
        kw["pos_embedding_strategy"] = "feature_map_based"
    elif variant == "hybrid":
        kw["pos_embedding_strategy"] = "hybrid"
    elif variant in ("vt_only_ln", "vt_only_fm_pos"):
        kw["use_vt_region_feature_only"] = True
        kw["region_feature_dim"] = 2048
        if variant == "vt_only_ln":
            kw["apply_region_layer_norm"] = True
        else:
            kw["pos_embedding_strategy"] = "feature_map_based"
    elif variant == "nopos":
        kw["apply_position_embedding"] = False
    elif variant == "aux_only":
        kw["use_vision_tower_region_feature"] = False
        kw["region_feature_dim"] = 3840
    torch.manual_seed(0)
    m = HFREModule(**kw)
    if variant in ("ln", "vt_only_ln"):
        p = ln_params()
        with torch.no_grad():           # the reference builds only the norms its configuration can reach
            if hasattr(m, "aux_region_norm"):
                m.aux_region_norm.weight.copy_(p["aux_w"]); m.aux_region_norm.bias.copy_(p["aux_b"])
            if hasattr(m, "vt_region_norm"):
                m.vt_region_norm.weight.copy_(p["vt_w"]); m.vt_region_norm.bias.copy_(p["vt_b"])

    class _Fixed(torch.nn.Module):
        def forward(self, x):
            return case["fpn_maps"]
    m.simple_fpn = _Fixed()
    gh, gw = case["grid_hw"]
    vt_in = torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16)
    with torch.no_grad():
        out = m(aux_multi_level_features=case["aux_maps"], aux_boxes=[case["boxes"].clone()],
                vt_multi_level_features=vt_in, vt_boxes=[case["vt_boxes"].clone()])
    return out.squeeze(0)


def main():
    case = make_case("demo_fpn")
    blobs = dict(checksum=checksum(case))
    for v in ("ln", "aux_pos", "vt_only", "fm_pos", "hybrid", "vt_only_ln", "vt_only_fm_pos", "nopos"):
        out = run(case, v)
        blobs[v] = out.numpy()
        print(v, tuple(out.shape))
    try:
        run(case, "aux_only")
        blobs["aux_only_error"] = np.array("none")
    except Exception as e:            # noqa: BLE001 — the exception type IS the datum
        blobs["aux_only_error"] = np.array(type(e).__name__)
    print("aux_only ->", blobs["aux_only_error"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hfre_variants.npz"), **blobs)


if __name__ == "__main__":
    main()
