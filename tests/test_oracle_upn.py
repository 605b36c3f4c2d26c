"""The torch restatement of the UPN deformable encoder (oracle/upn_oracle.py) against goldens made by the reference's own modules
(tests/golden/make_upn_golden.py)."""
import os

import numpy as np
import torch

import upn_cases as C
from oracle import upn_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "upn_ref.npz"))


def test_deformable_encoder_matches_reference_modules():
    state = C.encoder_state(2)
    src, pos = C.encoder_inputs()
    ref = O.encoder_reference_points(C.ENC_SHAPES)
    a0 = O.ms_deform_attn(state, "layers.0.self_attn.", src + pos, ref, src, C.ENC_SHAPES, C.level_start(C.ENC_SHAPES))
    assert (a0 - torch.from_numpy(G["enc.layer0.self_attn"])).abs().max().item() <= 2e-5
    outs = []
    mem = O.encoder(state, src, pos, C.ENC_SHAPES, 2, collect=outs)
    assert (outs[0] - torch.from_numpy(G["enc.layer0"])).abs().max().item() <= 5e-5
    assert (mem - torch.from_numpy(G["enc.memory"])).abs().max().item() <= 1e-4
