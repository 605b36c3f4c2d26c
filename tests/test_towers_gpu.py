"""GPU parity of the three vision sub-models (engine over the C-ABI) vs their CPU oracles at TRUE
channel widths.  Tolerance: per-token cosine >= 0.9995 (bf16 activations re-rounded at every
op through 8-24 residual blocks) and max|delta| / max|ref| <= 2^-4 for tower outputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def check(got, ref, what, cos_min=0.9995, rel_max=2 ** -4):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    rel = (got - ref).abs().max() / ref.abs().max()
    assert cos.min() >= cos_min, f"{what}: min cosine {cos.min():.6f} (token {int(cos.argmin())}), rel {rel:.4g}"
    assert rel <= rel_max, f"{what}: max rel err {rel:.4g}, min cos {cos.min():.6f}"


@pytest.mark.parametrize("gh,gw", [(34, 46), (28, 36), (10, 6)])
def test_vit_true_width(gh, gw):
    from oracle import vit_oracle as VO
    from vlm_fo1_amd.vit import QwenViT, ViTConfig
    depth, full = 4, (1, 3)
    sd = VO.random_vit_state(depth, 1280, 16, 3420, 2048, seed=gh)
    cfg = ViTConfig(depth=depth, fullatt_block_indexes=full)
    eng = QwenViT(cfg, sd, "cuda")
    g = torch.Generator().manual_seed(3)
    pix = torch.randn(gh * gw, 1176, generator=g).bfloat16()
    tokens, feats = eng.forward(pix.cuda(), gh, gw)
    ref_tokens, ref_maps = VO.vit_forward(sd, pix.float(), gh, gw, depth=depth, n_heads=16, fullatt=full)
    assert len(feats) == 2
    for i, (a, b) in enumerate(zip(feats, ref_maps)):
        check(a, b, f"vit {gh}x{gw} captured map {i}")
    check(tokens, ref_tokens, f"vit {gh}x{gw} image tokens")
    # capture="last" returns only the final full-attention map (SimpleFPN variant)
    _, last = eng.forward(pix.cuda(), gh, gw, capture="last")
    assert len(last) == 1 and torch.equal(last[0], feats[-1])


@pytest.mark.parametrize("H,W", [(480, 640), (399, 500)])
def test_davit_large(H, W):
    from oracle import davit_oracle as DO
    from vlm_fo1_amd.davit import DaViT
    sd = DO.random_davit_state(DO.DAVIT_LARGE, seed=1)
    eng = DaViT(sd, "cuda")
    g = torch.Generator().manual_seed(H)
    img = torch.randn(1, 3, H, W, generator=g).bfloat16()
    outs, sizes = eng.forward(img.cuda())
    ref, ref_sizes = DO.davit_forward(sd, img.float())
    assert sizes == ref_sizes
    # stage 3 sits behind 48 residual sub-blocks; the REFERENCE's own bf16 execution (torch CPU bf16,
    # same weights/input, measured in the build container) deviates from fp32 by min cosine 0.99936 /
    # rel 0.031 there and 0.99969 at stage 2 — the per-stage floors below are that intrinsic bf16 noise.
    floors = [0.9995, 0.9995, 0.9995, 0.9990]
    for i, (a, b) in enumerate(zip(outs, ref)):
        check(a, b, f"davit {H}x{W} stage {i}", cos_min=floors[i])


def test_simple_fpn_true_width():
    from oracle import fpn_oracle as FO
    from vlm_fo1_amd.fpn import SimpleFPN
    sd = FO.random_fpn_state(seed=4)
    eng = SimpleFPN(sd, "cuda")
    gh, gw = 34, 46
    g = torch.Generator().manual_seed(5)
    x = torch.randn(gh * gw, 1280, generator=g).bfloat16()
    outs, sizes = eng.forward(x.cuda(), gh, gw)
    ref = FO.fpn_forward(sd, x.float().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
    assert sizes == [(136, 184), (68, 92), (34, 46), (17, 23)]
    for i, (a, r) in enumerate(zip(outs, ref)):
        check(a, r[0].permute(1, 2, 0).reshape(-1, 512), f"fpn level {i}", cos_min=0.9998, rel_max=2 ** -5)
