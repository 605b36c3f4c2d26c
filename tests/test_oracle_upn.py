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


def test_query_selection_decoder_and_heads_match_reference_model():
    """oracle/upn_oracle.py query_selection + decoder against the reference's DeformableTransformer + UPN heads (2 + 2 layers,
    30 queries), fed with the reference's own encoder memory."""
    state = C.transformer_state(2, 2, C.N_QUERIES_SMALL)
    memory = torch.from_numpy(G["tr.memory"])[0]
    src, pos = C.encoder_inputs(seed=78)
    enc_state = {k[len("transformer.encoder."):]: v for k, v in state.items() if k.startswith("transformer.encoder.")}
    mem2 = O.encoder(enc_state, src, pos, C.ENC_SHAPES, 2)[0]
    assert (mem2 - memory).abs().max().item() <= 1e-4
    scores, coords, idx, refp = O.query_selection(state, memory, C.ENC_SHAPES, C.N_QUERIES_SMALL)
    assert (scores - torch.from_numpy(G["tr.sel.scores"])[0]).abs().max().item() <= 1e-4
    gc = torch.from_numpy(G["tr.sel.coords"])[0]
    fin = torch.isfinite(gc)
    assert torch.equal(fin, torch.isfinite(coords)) and (coords[fin] - gc[fin]).abs().max().item() <= 1e-4
    assert (refp - torch.from_numpy(G["tr.sel.refpoints"])[0]).abs().max().item() <= 1e-4
    hs, refs, boxes, logits = O.decoder(state, memory, C.ENC_SHAPES, torch.from_numpy(G["tr.sel.refpoints"])[0], 2)
    assert (hs - torch.from_numpy(G["tr.hs"])[:, 0]).abs().max().item() <= 2e-4
    assert (refs - torch.from_numpy(G["tr.refs"])[:, 0]).abs().max().item() <= 1e-5
    assert (boxes - torch.from_numpy(G["tr.pred_boxes"])[0]).abs().max().item() <= 1e-5
    assert (logits - torch.from_numpy(G["tr.pred_logits"])[0]).abs().max().item() <= 2e-4


def test_swin_backbone_input_projection_and_whole_detector_match_reference_model():
    """Swin-L widths at depths [2, 2, 2, 2] on a 100 x 136 image (ragged against the 4-pixel patch grid's windows: padding, cyclic
    shift and the shift mask are all exercised), input projections + position embeddings, and the whole forward down to the boxes."""
    state = C.upn_state()
    img = C.test_image()
    feats, sizes = O.swin_forward(state, img, C.SWIN_DEPTHS_SMALL, C.SWIN_HEADS, C.SWIN_WINDOW)
    assert sizes == [(25, 34), (13, 17), (7, 9), (4, 5)]
    for l, f in enumerate(feats):
        ref = torch.from_numpy(G[f"full.swin{l}"]).float()
        assert (f - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item()), f"swin stage {l}"      # golden stored in fp16
    src, pos, shapes = O.backbone_encoder_inputs(state, feats, sizes)
    assert [list(s) for s in shapes] == G["full.shapes"].tolist()
    assert (src - torch.from_numpy(G["full.src"]).float()).abs().max().item() <= 1e-2
    assert (pos - torch.from_numpy(G["full.pos"]).float()).abs().max().item() <= 5e-3
    boxes, logits = O.detect(state, img, C.SWIN_DEPTHS_SMALL, C.SWIN_HEADS, C.SWIN_WINDOW, 2, 2, C.N_QUERIES_SMALL)
    assert (boxes - torch.from_numpy(G["full.pred_boxes"])).abs().max().item() <= 1e-4
    assert (logits - torch.from_numpy(G["full.pred_logits"])).abs().max().item() <= 1e-3
