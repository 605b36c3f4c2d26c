"""encode_regions pinned to the reference's OWN method: tests/golden/encode_regions_ref.npz holds what
`OmChatQwen25VLForCausalLM.encode_regions` (omchat_qwen2_5_vl.py:75-128, run in place by tests/golden/make_encode_regions_golden.py
with the reference's HFREModule inside) returns on seeded cases — with and without SimpleFPN, boxes on the image border, and NO boxes
(the dummy box [0, 10, 0, 10] of :90-91).  Checked here: the oracle composition the full-depth parity test uses (vt-space box scaling,
hfre_oracle, cast to the tower dtype, projector)."""
import os

import numpy as np
import pytest
import torch

import encode_regions_cases as EC
from oracle import hfre_oracle as HO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encode_regions_ref.npz"))


@pytest.mark.parametrize("name", EC.NAMES)
def test_oracle_composition_equals_reference_encode_regions(name):
    c = EC.make(name)
    ref = torch.from_numpy(G[name])
    boxes = c["boxes_in"]
    if boxes is None or len(boxes) == 0:
        boxes = torch.tensor([[0, 10, 0, 10]], dtype=torch.float32)             # :90-91
    boxes = boxes.to(torch.float32)
    H, W = c["img"]
    gh, gw = c["grid_hw"]
    # :94-99 — (primary tower input size) / (aux tensor size) per axis, in tensor arithmetic
    sh = torch.tensor(gh * 14) / H
    sw = torch.tensor(gw * 14) / W
    vt_boxes = boxes * torch.tensor([sw, sh, sw, sh])
    vt = c["fpn_maps"] if c["fpn"] else c["vt_maps"]
    feat = HO.hfre_oracle(c["aux_maps"], boxes, vt, vt_boxes, region_dim=c["region_dim"], grid_hw=(gh, gw),
                          vt_strides=[3.5, 7, 14, 28] if c["fpn"] else None)[0]
    w, b = EC.projector(c["region_dim"])
    tok = torch.nn.functional.linear(feat.to(torch.bfloat16), w, b).float()      # :106-107: cast to the tower dtype, then mm_projector_aux
    assert tok.shape == ref.shape
    # the HFRE oracle equals the reference module to ~1e-6 (tests/test_oracle_hfre.py): a bf16 cast may flip on a handful of features
    torch.testing.assert_close(tok, ref, rtol=2 ** -6, atol=2e-3)
    assert (tok == ref).float().mean() > 0.97
