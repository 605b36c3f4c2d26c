"""Ragged image batches: DaViT-L and SimpleFPN over images of DIFFERENT sizes in ONE pass — rows packed image by image,
the spatial kernels (depthwise conv + LayerNorm, im2col, window partition / reverse, channel attention, pixel shuffle, max-pool) reading
a per-image geometry table (include/fo1.h `fo1_img_seg`).  The reference runs its towers image by image (davit_aux_encoder.py:54-69,
simple_fpn.py:100-216 per call); here the packed pass must give, for every image, EXACTLY what its own one-image pass gives: with the GEMM
tile pinned (the k-order of a row's dot products then does not depend on M) every map is bit-identical."""
import pytest
import torch

# the whole module compares pinned-tile passes bit for bit: it runs on the test / bench build (conftest.ab_library_module)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ab_library_module")]

SIZES = [(399, 500), (711, 333), (60, 64), (420, 420), (97, 233), (480, 640)]      # incl. sizes that pad the 12 x 12 windows and odd extents


def pinned():
    from vlm_fo1_amd import lib as L

    class _Pin:
        def __enter__(self):
            L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")
            L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
            L.check(L.load().fo1_gemm_set_gemv(0), "gemv")

        def __exit__(self, *exc):
            L.load().fo1_gemm_set_variant(0, 0)
            L.load().fo1_gemm_set_splitk(0)
            L.load().fo1_gemm_set_gemv(1)
            return False
    return _Pin()


@pytest.fixture(scope="module")
def towers():
    from vlm_fo1_amd.davit import DaViT
    from vlm_fo1_amd.fpn import SimpleFPN
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config, random_weights
    from vlm_fo1_amd.vit import ViTConfig
    cfg = FO1Config(vit=ViTConfig(depth=1, fullatt_block_indexes=(0,)), llm=LLMConfig(num_layers=1, vocab_size=1024, max_seq=256))
    W = random_weights(cfg, "cuda", seed=11)
    return DaViT(W["davit"], "cuda"), SimpleFPN(W["fpn"], "cuda")


def test_davit_ragged_pass_equals_one_image_passes(towers):
    davit, _ = towers
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randn(3, H, W, generator=g).bfloat16().cuda() for H, W in SIZES]
    with pinned():
        single = []
        for im in imgs:
            maps, sizes = davit.forward(im)
            single.append(([m.clone() for m in maps], sizes))
        maps, plan = davit.forward_ragged(imgs)
        torch.cuda.synchronize()
    for b, (ref, sizes) in enumerate(single):
        for l in range(4):
            assert plan.sizes[l][b] == sizes[l]
            h, w = sizes[l]
            r0 = plan.row0[l][b]
            got = maps[l][r0:r0 + h * w]
            assert torch.equal(got, ref[l]), f"image {b} ({SIZES[b]}) level {l}: ragged pass differs from the one-image pass"
    # a different order permutes the rows, nothing else
    perm = [3, 0, 5, 1, 4, 2]
    with pinned():
        maps2, plan2 = davit.forward_ragged([imgs[j] for j in perm])
        torch.cuda.synchronize()
    for slot, j in enumerate(perm):
        for l in range(4):
            h, w = plan2.sizes[l][slot]
            r0 = plan2.row0[l][slot]
            assert torch.equal(maps2[l][r0:r0 + h * w], single[j][0][l])


def test_fpn_ragged_pass_equals_one_image_passes(towers):
    _, fpn = towers
    grids = [(28, 36), (50, 24), (4, 4), (30, 30), (6, 16), (34, 46)]
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(gh * gw, 1280, generator=g).bfloat16().cuda() for gh, gw in grids]
    row0, off = [], 0
    for x in xs:
        row0.append(off)
        off += x.shape[0]
    with pinned():
        single = []
        for x, (gh, gw) in zip(xs, grids):
            maps, sizes = fpn.forward(x, gh, gw)
            single.append(([m.clone() for m in maps], sizes))
        maps, plan = fpn.forward_ragged(torch.cat(xs, 0), grids, row0)
        torch.cuda.synchronize()
    for b, (ref, sizes) in enumerate(single):
        for l in range(4):
            assert tuple(plan.sizes[l][b]) == tuple(sizes[l])
            h, w = sizes[l]
            r0 = plan.row0[l][b]
            assert torch.equal(maps[l][r0:r0 + h * w], ref[l]), f"image {b} (grid {grids[b]}) level {l}: ragged FPN differs from the one-image pass"


def test_ragged_passes_take_the_implicit_convolution_bitwise(towers, monkeypatch):
    """Ragged packs big enough for the 256 x 256 GEMM kernel (unpinned): DaViT's pre-norm ConvEmbed of stage 1 and the 3x3 output convolutions of
    the finer FPN levels run as implicit GEMMs over ONE zero-framed buffer with a common row pitch (ops.Conv3x3Plan over images of different
    sizes) — the same bits as layernorm + im2col_var + gemm (FO1_CONV_IMPLICIT=0)."""
    from vlm_fo1_amd import ops
    davit, fpn = towers
    g = torch.Generator().manual_seed(8)
    sizes = [(480, 640), (399, 500), (640, 480), (420, 420), (333, 711), (480, 640), (512, 384)]
    imgs = [torch.randn(3, H, W, generator=g).bfloat16().cuda() for H, W in sizes]
    grids = [(34, 46), (28, 36), (46, 34), (30, 30), (24, 50), (34, 46)]
    xs = torch.cat([torch.randn(gh * gw, 1280, generator=g).bfloat16().cuda() for gh, gw in grids], 0)
    row0, off = [], 0
    for gh, gw in grids:
        row0.append(off)
        off += gh * gw
    lv0 = [((H + 2 * 3 - 7) // 4 + 1, (W + 2 * 3 - 7) // 4 + 1) for H, W in sizes]                   # DaViT stage-0 maps = the input of stage 1's 3x3 / stride 2 embed
    assert ops.conv3x3_implicit_for(lv0, 2, 512, 256, 3, 1), "the test's pack must be large enough for the implicit form"
    assert ops.conv3x3_implicit_for([(4 * a, 4 * b) for a, b in grids], 1, 512, 512, 3, 1)

    def run():
        m, _ = davit.forward_ragged(imgs)
        f, _ = fpn.forward_ragged(xs, grids, row0)
        torch.cuda.synchronize()
        return [t.clone() for t in m], [t.clone() for t in f]

    m1, f1 = run()
    monkeypatch.setenv("FO1_CONV_IMPLICIT", "0")
    m0, f0 = run()
    for l, (a, b) in enumerate(zip(m1, m0)):
        assert torch.equal(a, b), f"DaViT level {l}: implicit vs im2col_var"
    for l, (a, b) in enumerate(zip(f1, f0)):
        assert torch.equal(a, b), f"FPN level {l}: implicit vs im2col_var"


def test_ragged_spatial_kernels_against_torch_references():
    """The geometry-table path of each spatial kernel against plain torch on two images of different sizes (the same-size path has its
    own references in tests/test_vision_ops_gpu.py; here: no cross-image reads, right offsets, per-image channel-attention scale)."""
    import torch.nn.functional as F
    from vlm_fo1_amd import ops
    sizes = [(13, 17), (25, 12)]
    C = 64
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(h * w, C, generator=g).bfloat16() for h, w in sizes]
    row0 = [0, sizes[0][0] * sizes[0][1]]
    npx = [h * w for h, w in sizes]
    x = torch.cat(xs).cuda()
    # max-pool
    out_sizes = [(h // 2, w // 2) for h, w in sizes]
    o0 = [0, out_sizes[0][0] * out_sizes[0][1]]
    sg = ops.ImgSegs([(r, h, w, o, a, b) for r, (h, w), o, (a, b) in zip(row0, sizes, o0, out_sizes)], "cuda", max(npx), sum(npx),
                     max(a * b for a, b in out_sizes), sum(a * b for a, b in out_sizes))
    y = ops.maxpool2_var(x, sg).cpu()
    for b, (h, w) in enumerate(sizes):
        ref = F.max_pool2d(xs[b].float().reshape(h, w, C).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0).reshape(-1, C).bfloat16()
        a, bb = out_sizes[b]
        assert torch.equal(y[o0[b]:o0[b] + a * bb], ref)
    # channel attention: per-image softmax((q N^-1/2)^T k) per 32-channel group
    qkv = torch.randn(sum(npx), 3 * C, generator=g).bfloat16()
    tok = ops.ImgSegs([(r, n) for r, n in zip(row0, npx)], "cuda", max(npx), sum(npx), max(npx), sum(npx))
    got = ops.channel_attention_var(qkv.cuda(), C, tok).cpu()
    for b, n in enumerate(npx):
        one = ops.channel_attention(qkv[row0[b]:row0[b] + n].cuda().contiguous(), C).cpu()
        assert torch.equal(got[row0[b]:row0[b] + n], one), f"channel attention image {b}: table path differs from the one-image call"
