"""GPU parity tests of the dense / elementwise / attention ops (through the C-ABI) against plain
torch fp32 references of the same op on the CPU, with the reference's bf16 rounding points.

Tolerances: an op that ends in ONE bf16 rounding of an fp32 result must match the reference's
rounding to <= 1 bf16 ulp (rtol 2^-7 on the value); accumulations (GEMM, attention) use
rtol 1.6e-2 / atol scaled to the output magnitude (SURVEY §7 'parity tolerances')."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(x):
    return x.to(BF).float()


def close_bf16(got, ref, ulps=1.0, atol=1e-6, what="", rare=0.0):
    """|got - ref| <= ulps bf16 ulps (+ atol) everywhere; `rare` > 0 lets that fraction of the elements be up to twice as far:
    a result that passes through TWO bf16 roundings (silu -> bf16 -> * up -> bf16) moves two ulps when an fp32 difference of a
    few ulps (v_exp_f32 / v_rcp_f32 vs libm) flips the first rounding — expected about once per 1e4-1e5 elements."""
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = ulps * 2.0 ** -7 * ref.abs() + atol
    bad = (got - ref).abs() > tol
    if rare > 0.0 and bad.any():
        assert bad.float().mean().item() <= rare, f"{what}: {int(bad.sum())}/{bad.numel()} beyond {ulps} ulps (allowed fraction {rare})"
        bad = (got - ref).abs() > 2 * tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max abs err {(got - ref).abs().max():.4g} " \
                          f"at ref {ref.flatten()[(got - ref).abs().argmax()]:.4g}"


def gemm_ref(a, w, bias=None, res=None, act=0):
    y = a.float().cpu() @ w.float().cpu().t()
    if bias is not None:
        y = y + bias.float().cpu()
    y = rb(y)
    if act == 1:
        y = rb(torch.nn.functional.gelu(y))
    elif act == 2:
        y = rb(torch.nn.functional.silu(y))
    if res is not None:
        y = rb(y + res.float().cpu())
    return y


GEMM_SHAPES = [
    # (M, N, K, bias, res, act)
    (64, 64, 64, False, False, 0),
    (128, 256, 128, True, False, 0),
    (200, 328, 1176, True, False, 0),      # patch-embed K=1176 (K % 64 != 0 -> register path), ragged M/N
    (391, 2560, 2048, True, False, 0),     # LLM qkv
    (391, 2048, 11008, False, True, 0),    # LLM down + residual
    (100, 2048, 5888, True, False, 1),     # region projector + GELU
    (1564, 3424, 1280, True, False, 0),    # ViT MLP (padded 3420 -> 3424)
    (77, 130, 72, True, True, 2),          # tiny ragged everything
]


@pytest.mark.parametrize("staging", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_gemm_variants(staging, tile, ab_library):
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(staging * 10 + tile)
    try:
        for (M, N, K, hb, hr, act) in GEMM_SHAPES:
            if staging >= 2 and K % 64 != 0:
                continue
            L.check(L.load().fo1_gemm_set_variant(staging, tile), "variant")
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            res = torch.randn(M, N).to(BF).cuda() if hr else None
            got = ops.gemm(a, w, bias, res, act)
            ref = gemm_ref(a, w, bias, res, act)
            scale = ref.abs().max().item()
            err = (got.float().cpu() - ref).abs().max().item()
            assert err <= 2e-2 * scale + 1e-3, f"gemm {M}x{N}x{K} staging={staging} tile={tile}: max err {err:.4g} (scale {scale:.4g})"
            # transposition / tile-mapping sanity: relative Frobenius error must be at rounding level
            rel = (got.float().cpu() - ref).norm() / ref.norm()
            assert rel < 4e-3, f"gemm {M}x{N}x{K} staging={staging} tile={tile}: rel fro err {rel:.4g}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)


def test_gemm_auto_and_fp32_out_and_strided():
    from vlm_fo1_amd import ops
    torch.manual_seed(5)
    M, N, K = 300, 512, 256
    big = (torch.randn(M, K + 64) * 0.5).to(BF).cuda()
    a = big[:, 32:32 + K]  # row-strided view, 16-B aligned offset (32*2 B)
    w = (torch.randn(N, K) * 0.05).to(BF).cuda()
    got = ops.gemm(a, w, out_f32=True)
    ref = a.float().cpu() @ w.float().cpu().t()
    assert got.dtype == torch.float32
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-3)
    # asymmetric check: A = I picks rows of W^T exactly (catches row/col swaps)
    eye = torch.eye(128, dtype=BF).cuda()
    w2 = torch.arange(96 * 128, dtype=torch.float32).reshape(96, 128).remainder(251).to(BF).cuda()
    got2 = ops.gemm(eye, w2)
    assert torch.equal(got2.cpu(), w2.cpu().t().contiguous())


def test_rmsnorm_layernorm():
    from vlm_fo1_amd import ops
    torch.manual_seed(1)
    for D in (1280, 2048, 256, 512):
        x = (torch.randn(37, D) * 3).to(BF).cuda()
        w = (1 + 0.1 * torch.randn(D)).to(BF).cuda()
        b = (0.1 * torch.randn(D)).to(BF).cuda()
        got = ops.rmsnorm(x, w, 1e-6)
        xf = x.float().cpu()
        ref = rb(w.float().cpu() * rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)))
        close_bf16(got, ref, ulps=1.01, what=f"rmsnorm D={D}")
        got = ops.layernorm(x, w, b, 1e-5)
        ref = rb(torch.nn.functional.layer_norm(xf, (D,), w.float().cpu(), b.float().cpu(), 1e-5))
        close_bf16(got, ref, ulps=1.01, atol=2e-3, what=f"layernorm D={D}")


def test_swiglu_biasact_argmax():
    from vlm_fo1_amd import ops
    torch.manual_seed(2)
    gu = (torch.randn(33, 2 * 3424) * 2).to(BF).cuda()
    got = ops.swiglu(gu)
    g, u = gu.float().cpu()[:, :3424], gu.float().cpu()[:, 3424:]
    ref = rb(rb(torch.nn.functional.silu(g)) * u)
    close_bf16(got, ref, ulps=1.01, atol=1e-5, what="swiglu", rare=1e-4)
    x = torch.randn(50, 640).to(BF).cuda()
    bias = torch.randn(640).to(BF).cuda()
    got = ops.bias_act(x, bias, 1)
    ref = rb(torch.nn.functional.gelu(rb(x.float().cpu() + bias.float().cpu())))
    close_bf16(got, ref, ulps=1.01, atol=1e-5, what="bias+gelu", rare=1e-4)
    row = torch.randn(151936).to(BF)
    row[777] = row.max() + 1
    row[90000] = row[777]  # tie: first index wins
    assert ops.argmax(row.cuda()).item() == 777


def test_rope_llm_and_kcache():
    from vlm_fo1_amd import ops
    torch.manual_seed(3)
    L, H, KV, HD = 45, 16, 2, 128
    qkv = torch.randn(L, (H + 2 * KV) * HD).to(BF).cuda()
    ang = torch.rand(L, HD // 2) * 6.28
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(BF), emb.sin().to(BF)
    ref_in = qkv.float().cpu()[:, :(H + KV) * HD].reshape(L, H + KV, HD)
    c, s = cos.float()[:, None, :], sin.float()[:, None, :]
    rot = torch.cat([-ref_in[..., HD // 2:], ref_in[..., :HD // 2]], -1)
    ref = rb(rb(ref_in * c) + rb(rot * s))
    kc = torch.zeros(KV, 64, HD, dtype=BF, device="cuda")
    v_before = qkv[:, (H + KV) * HD:].clone()
    ops.rope_llm(qkv, H + KV, HD, cos.cuda(), sin.cuda(), kcache=kc, k_first_head=H, pos0=7)
    got = qkv.float().cpu()[:, :(H + KV) * HD].reshape(L, H + KV, HD)
    close_bf16(got, ref, ulps=1.01, atol=1e-6, what="rope_llm")
    assert torch.equal(qkv[:, (H + KV) * HD:], v_before), "v must be untouched"
    assert torch.equal(kc[:, 7:7 + L].cpu(), qkv[:, H * HD:(H + KV) * HD].reshape(L, KV, HD).permute(1, 0, 2).cpu())
    assert kc[:, :7].abs().sum() == 0 and kc[:, 7 + L:].abs().sum() == 0


def test_rope_vit():
    from vlm_fo1_amd import ops
    torch.manual_seed(4)
    S, H, HD = 100, 16, 80
    qkv = torch.randn(S, 3 * H * HD).to(BF).cuda()
    ang = torch.rand(S, HD // 2) * 6.28
    x = qkv.float().cpu()[:, :2 * H * HD].reshape(S, 2 * H, HD)
    emb = torch.cat([ang, ang], -1)
    c, s = emb.cos()[:, None, :], emb.sin()[:, None, :]
    rot = torch.cat([-x[..., HD // 2:], x[..., :HD // 2]], -1)
    ref = rb(x * c + rot * s)
    v_before = qkv[:, 2 * H * HD:].clone()
    ops.rope_vit(qkv, H, HD, ang.cos().cuda(), ang.sin().cuda())
    got = qkv.float().cpu()[:, :2 * H * HD].reshape(S, 2 * H, HD)
    close_bf16(got, ref, ulps=1.01, atol=1e-6, what="rope_vit")
    assert torch.equal(qkv[:, 2 * H * HD:], v_before)


def test_transpose():
    from vlm_fo1_amd import ops
    torch.manual_seed(6)
    for (M, C, col0, ldd) in [(100, 128, 0, 128), (37, 256, 8, 72), (391, 1280, 0, 448), (5, 64, 3, 16)]:
        src = torch.randn(M, C).to(BF).cuda()
        dst = torch.zeros(C, ldd, dtype=BF, device="cuda")
        ops.transpose_into(src, dst, col0)
        ref = torch.zeros(C, ldd, dtype=BF)
        ref[:, col0:col0 + M] = src.cpu().t()
        assert torch.equal(dst.cpu(), ref), f"transpose {M}x{C} col0={col0}"


def attn_ref(q, k, v, segments, causal, scale):
    """q [L,H,D], k/v [L,KV,D] fp32 cpu; P rounded to bf16 before PV like the kernel / flash-attn."""
    L, H, D = q.shape
    KV = k.shape[1]
    out = torch.zeros(L, H, D)
    for (s, e) in segments:
        for h in range(H):
            kv = h // (H // KV)
            sc = (q[s:e, h] @ k[s:e, kv].t()) * scale
            if causal:
                mask = torch.ones(e - s, e - s, dtype=torch.bool).tril()
                sc = sc.masked_fill(~mask, float("-inf"))
            m = sc.max(-1, keepdim=True).values
            p = rb(torch.exp(sc - m))
            out[s:e, h] = (p @ v[s:e, kv]) / p.sum(-1, keepdim=True)
    return out


@pytest.mark.parametrize("cfg", [
    dict(name="llm_causal_gqa", L=391, H=16, KV=2, D=128, segs=None, causal=True),
    dict(name="llm_causal_short", L=50, H=16, KV=2, D=128, segs=None, causal=True),
    dict(name="vit_full", L=300, H=16, KV=16, D=80, segs=None, causal=False),
    dict(name="vit_windows_ragged", L=296, H=16, KV=16, D=80, segs=[(0, 64), (64, 128), (128, 160), (160, 176), (176, 240), (240, 296)], causal=False),
    dict(name="davit_window", L=288, H=8, KV=8, D=32, segs=[(0, 144), (144, 288)], causal=False),
])
def test_attention(cfg):
    from vlm_fo1_amd import ops
    torch.manual_seed(7)
    L, H, KV, D = cfg["L"], cfg["H"], cfg["KV"], cfg["D"]
    segs = cfg["segs"] or [(0, L)]
    # fused qkv buffer like the engine uses it
    qkv = (torch.randn(L, (H + 2 * KV) * D) * 1.5).to(BF).cuda()
    q = qkv[:, :H * D]
    k = qkv[:, H * D:(H + KV) * D]
    v = qkv[:, (H + KV) * D:]
    Lp = (L + 63) // 64 * 64
    vt = torch.zeros(KV * D, Lp, dtype=BF, device="cuda")
    ops.transpose_into(v, vt, 0)
    scale = 1.0 / math.sqrt(D)
    ref = attn_ref(q.float().cpu().reshape(L, H, D), k.float().cpu().reshape(L, KV, D), v.float().cpu().reshape(L, KV, D),
                   segs, cfg["causal"], scale).reshape(L, H * D)
    outs = []
    for blk in (64, 32, 16):   # 4 / 2 / 1 waves per workgroup
        items = ops.make_items(segs, "cuda", block=blk)
        got = ops.attention(q, k, vt, items, H, KV, D, scale, cfg["causal"])
        err = (got.float().cpu() - ref).abs()
        assert err.max() < 3e-2, f"{cfg['name']} q_block={blk}: max err {err.max():.4g}; worst row {int(err.max(1).values.argmax())}"
        rel = (got.float().cpu() - ref).norm() / ref.norm()
        assert rel < 6e-3, f"{cfg['name']} q_block={blk}: rel fro err {rel:.4g}"
        outs.append(got)
    # the query-block size is a pure work partition: identical arithmetic per query
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a later key tile (spiked K row): the rescale of O and l
    must be exact (guide rule 26: a rare data-dependent branch needs its own test)."""
    from vlm_fo1_amd import ops
    torch.manual_seed(8)
    L, H, D = 256, 2, 128
    q = torch.randn(L, H * D)
    k = torch.randn(L, H * D) * 0.3
    v = torch.randn(L, H * D)
    k[200] = q[10] * 2.0  # key 200 (4th tile) dominates query 10
    k[70] = q[33] * 3.0
    qkv = torch.cat([q, k, v], 1).to(BF).cuda()
    vt = torch.zeros(H * D, 256, dtype=BF, device="cuda")
    ops.transpose_into(qkv[:, 2 * H * D:], vt, 0)
    items = ops.make_items([(0, L)], "cuda")
    sc = 1 / math.sqrt(D)
    got = ops.attention(qkv[:, :H * D], qkv[:, H * D:2 * H * D], vt, items, H, H, D, sc, False)
    qf = qkv.float().cpu()
    ref = attn_ref(qf[:, :H * D].reshape(L, H, D), qf[:, H * D:2 * H * D].reshape(L, H, D),
                   qf[:, 2 * H * D:].reshape(L, H, D), [(0, L)], False, sc).reshape(L, H * D)
    err = (got.float().cpu() - ref).abs()
    assert err.max() < 3e-2, f"max err {err.max():.4g} at row {int(err.max(1).values.argmax())}"


def _attn_case(segs, H, KV, D, seed):
    from vlm_fo1_amd import ops
    torch.manual_seed(seed)
    L = max(e for _, e in segs)
    qkv = (torch.randn(L, (H + 2 * KV) * D) * 1.5).to(BF).cuda()
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    vt = torch.zeros(KV * D, (L + 63) // 64 * 64, dtype=BF, device="cuda")
    ops.transpose_into(v, vt, 0)
    return qkv, q, k, v, vt, L


@pytest.mark.parametrize("cfg", [
    dict(name="llm_packed_gqa", H=16, KV=2, D=128, segs=[(0, 652), (652, 1304), (1304, 1364), (1364, 1500)], causal=True, blk=128),
    dict(name="llm_one_sequence", H=16, KV=2, D=128, segs=[(0, 391)], causal=True, blk=128),
    dict(name="vit_full_two_images", H=16, KV=16, D=80, segs=[(0, 1564), (1564, 1864)], causal=False, blk=256),
    dict(name="hd128_mha_causal", H=4, KV=4, D=128, segs=[(0, 700)], causal=True, blk=256),
    dict(name="hd80_gqa_pairs", H=8, KV=4, D=80, segs=[(0, 300), (300, 556)], causal=False, blk=128),
])
def test_attention_32x32_form(cfg):
    """attn_fwd32_kernel (q_block 128 / 256: 32x32 MFMA, 8 waves x 32 queries, two query heads of one KV head per workgroup when
    grouped) against the fp32 reference at the 16x16 kernel's tolerance, and against the 16x16 kernel itself: the same products
    in another summation order (fp32 accumulate -> one bf16 rounding): equal to <= 2 bf16 ulps + 2e-3."""
    from vlm_fo1_amd import ops
    H, KV, D, segs = cfg["H"], cfg["KV"], cfg["D"], cfg["segs"]
    qkv, q, k, v, vt, L = _attn_case(segs, H, KV, D, 17)
    scale = 1.0 / math.sqrt(D)
    ref = attn_ref(q.float().cpu().reshape(L, H, D), k.float().cpu().reshape(L, KV, D), v.float().cpu().reshape(L, KV, D),
                   segs, cfg["causal"], scale).reshape(L, H * D)
    old = ops.attention(q, k, vt, ops.make_items(segs, "cuda", block=64), H, KV, D, scale, cfg["causal"])
    got = ops.attention(q, k, vt, ops.make_items(segs, "cuda", block=cfg["blk"]), H, KV, D, scale, cfg["causal"])
    err = (got.float().cpu() - ref).abs()
    assert err.max() < 3e-2, f"{cfg['name']}: max err {err.max():.4g}; worst row {int(err.max(1).values.argmax())}"
    rel = (got.float().cpu() - ref).norm() / ref.norm()
    assert rel < 6e-3, f"{cfg['name']}: rel fro err {rel:.4g}"
    d = (got.float() - old.float()).abs().cpu()
    assert (d <= 2 * 2.0 ** -7 * old.float().abs().cpu() + 2e-3).all(), f"{cfg['name']}: vs the 16x16 kernel max {d.max():.4g}"
    # run to run and partition to partition: a query's arithmetic depends on its segment and its place in the 32-query wave only
    again = ops.attention(q, k, vt, ops.make_items(segs, "cuda", block=cfg["blk"]), H, KV, D, scale, cfg["causal"])
    assert torch.equal(got, again)


def test_attention_32x32_form_prefix_ranges():
    """Second key range per item (shared prompt prefix, fo1_attention_prefix_bf16) on the 32x32 kernel: rows [0, 408) are the prefix,
    two prompts own [408, 660) and [660, 1000); every own row attends [prefix | own rows up to itself] = the causal attention of the
    sequence [prefix | own rows] restricted to its own rows."""
    from vlm_fo1_amd import ops
    H, KV, D = 16, 2, 128
    P, A, B = 408, 252, 340
    segs_all = [(0, P + A + B)]
    qkv, q, k, v, vt, L = _attn_case(segs_all, H, KV, D, 23)
    kc = k.reshape(L, KV, D).permute(1, 0, 2).contiguous()          # cache layout [kv head][pos][D]
    scale = 1.0 / math.sqrt(D)
    qf, kf, vf = q.float().cpu().reshape(L, H, D), k.float().cpu().reshape(L, KV, D), v.float().cpu().reshape(L, KV, D)
    ref = torch.zeros(L, H * D)
    ref[:P] = attn_ref(qf[:P], kf[:P], vf[:P], [(0, P)], True, scale).reshape(P, H * D)
    for (a, b) in ((P, P + A), (P + A, L)):
        idx = list(range(P)) + list(range(a, b))
        r = attn_ref(qf[idx], kf[idx], vf[idx], [(0, len(idx))], True, scale).reshape(len(idx), H * D)
        ref[a:b] = r[P:]
    for blk in (64, 128):
        rows, rng = [], []
        for (a, b, r2) in ((0, P, (0, 0)), (P, P + A, (0, P)), (P + A, L, (0, P))):
            for q0 in range(a, b, blk):
                rows.append([q0, min(q0 + blk, b), a, b]); rng.append(list(r2))
        items = torch.tensor(rows, dtype=torch.int32).cuda(); items.q_block = blk
        got = ops.attention_strided(q, 0, kc, vt, items, H, KV, D, scale, True, prefix_ranges=torch.tensor(rng, dtype=torch.int32).cuda())
        err = (got.float().cpu() - ref).abs()
        assert err.max() < 3e-2, f"q_block {blk}: max err {err.max():.4g}; worst row {int(err.max(1).values.argmax())}"
        assert (got.float().cpu() - ref).norm() / ref.norm() < 6e-3


def test_attention_32x32_form_rescale_branch():
    """The spiked-key case of test_attention_online_softmax_rescale_branch on the 32x32 kernel (the O rescale is a wave-uniform
    branch that random data takes only in the first tiles)."""
    from vlm_fo1_amd import ops
    torch.manual_seed(8)
    L, H, D = 512, 2, 128
    q = torch.randn(L, H * D)
    k = torch.randn(L, H * D) * 0.3
    v = torch.randn(L, H * D)
    k[200] = q[10] * 2.0
    k[70] = q[33] * 3.0
    k[450, :D] = q[300, :D] * 2.5
    qkv = torch.cat([q, k, v], 1).to(BF).cuda()
    vt = torch.zeros(H * D, 512, dtype=BF, device="cuda")
    ops.transpose_into(qkv[:, 2 * H * D:], vt, 0)
    sc = 1 / math.sqrt(D)
    qf = qkv.float().cpu()
    ref = attn_ref(qf[:, :H * D].reshape(L, H, D), qf[:, H * D:2 * H * D].reshape(L, H, D),
                   qf[:, 2 * H * D:].reshape(L, H, D), [(0, L)], False, sc).reshape(L, H * D)
    got = ops.attention(qkv[:, :H * D], qkv[:, H * D:2 * H * D], vt, ops.make_items([(0, L)], "cuda", block=256), H, H, D, sc, False)
    err = (got.float().cpu() - ref).abs()
    assert err.max() < 3e-2, f"q_block 256: max err {err.max():.4g} at row {int(err.max(1).values.argmax())}"
    # q_block 128 = two query heads per KV head: the same K / V under both query heads (KV head 0 <- k / v of head 0)
    k1, vt1 = qkv[:, H * D:H * D + D], vt[:D]
    got = ops.attention(qkv[:, :H * D], k1, vt1, ops.make_items([(0, L)], "cuda", block=128), H, 1, D, sc, False)
    ref1 = attn_ref(qf[:, :H * D].reshape(L, H, D), qf[:, H * D:H * D + D].reshape(L, 1, D), qf[:, 2 * H * D:2 * H * D + D].reshape(L, 1, D),
                    [(0, L)], False, sc).reshape(L, H * D)
    err = (got.float().cpu() - ref1).abs()
    assert err.max() < 3e-2, f"q_block 128: max err {err.max():.4g} at row {int(err.max(1).values.argmax())}"


@pytest.mark.parametrize("splits", [2, 3, 8])
def test_gemm_splitk(splits, ab_library):
    """Split-K partials + fixed-order reduce must match the single-pass kernel to fp32 re-association."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(11)
    try:
        for (M, N, K, hb, hr, act) in [(515, 2048, 11008, False, True, 0), (515, 2560, 2048, True, False, 0),
                                       (100, 2048, 5888, True, False, 1), (33, 132, 512, True, True, 2)]:
            if K % 64 != 0:
                continue
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            res = torch.randn(M, N).to(BF).cuda() if hr else None
            for tile in (2, 3):
                L.check(L.load().fo1_gemm_set_variant(2, tile), "variant")
                L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
                one = ops.gemm(a, w, bias, res, act).float().cpu()
                L.check(L.load().fo1_gemm_set_splitk(splits), "splitk")
                got = ops.gemm(a, w, bias, res, act).float().cpu()
                ref = gemm_ref(a, w, bias, res, act)
                scale = ref.abs().max().item()
                assert (got - ref).abs().max() <= 2e-2 * scale + 1e-3, f"splitk={splits} {M}x{N}x{K} tile={tile}"
                # vs the unsplit kernel: differences only from fp32 summation order (<= 1 bf16 ulp on a few elements)
                assert (got - one).abs().max() <= 2 ** -7 * scale + 1e-3
                got32 = ops.gemm(a, w, bias, None, 0, out_f32=True).cpu()
                ref32 = a.float().cpu() @ w.float().cpu().t() + (bias.float().cpu() if hb else 0)
                torch.testing.assert_close(got32, ref32, rtol=2e-3, atol=2e-3 * scale)
    finally:
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_variant(0, 0)


def test_gemm_fused_swiglu_epilogue(ab_library):
    """gate/up GEMM with SwiGLU in the epilogue == separate GEMM + swiglu kernel, bit for bit (same rounding points)."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(12)
    try:
        for (M, F, K, hb) in [(515, 11008, 2048, False), (1564, 3456, 1280, True), (37, 64, 128, True)]:
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            wg = (torch.randn(F, K) * 0.05).to(BF)
            wu = (torch.randn(F, K) * 0.05).to(BF)
            bg = (torch.randn(F) * 0.1).to(BF) if hb else None
            bu = (torch.randn(F) * 0.1).to(BF) if hb else None
            for tile in (0, 1, 2, 3):
                L.check(L.load().fo1_gemm_set_variant(0, tile), "variant")
                fused = ops.gemm(a, ops.interleave_gate_up(wg, wu).cuda(), ops.interleave_gate_up(bg, bu).cuda() if hb else None,
                                 act=ops.ACT_SWIGLU16)
                gu = ops.gemm(a, torch.cat([wg, wu], 0).cuda(), torch.cat([bg, bu], 0).cuda() if hb else None)
                ref = ops.swiglu(gu)
                assert fused.shape == (M, F)
                assert torch.equal(fused, ref), f"fused swiglu differs (M={M} F={F} tile={tile}): max {(fused.float() - ref.float()).abs().max()}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)


@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_gemv_matches_tile_gemm(M, ab_library):
    """The decode-step GEMV (M <= 4) must agree with the MFMA tile kernel on every epilogue form."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(20 + M)
    try:
        for (N, K, hb, hr, act) in [(2560, 2048, True, False, 0), (2048, 11008, False, True, 0), (151936, 2048, False, False, 0),
                                    (2048, 2048, False, True, 0), (22016, 2048, False, False, 3), (130, 72, True, True, 1),
                                    (6912, 1280, True, False, 3)]:
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            n_out = N // 2 if act == 3 else N
            res = torch.randn(M, n_out).to(BF).cuda() if hr else None
            L.load().fo1_gemm_set_gemv(1)
            got = ops.gemm(a, w, bias, res, act).float().cpu()
            L.load().fo1_gemm_set_gemv(0)
            ref = ops.gemm(a, w, bias, res, act).float().cpu()
            scale = ref.abs().max().item()
            err = (got - ref).abs().max().item()
            assert got.shape == ref.shape and err <= 2 ** -7 * scale + 1e-3, f"gemv M={M} N={N} K={K} act={act}: err {err:.4g} scale {scale:.4g}"
    finally:
        L.load().fo1_gemm_set_gemv(1)


def test_attention_decode_split_kv():
    """Flash-decoding path (KV split over workgroups, device-side kv length) == the single-workgroup kernel."""
    from vlm_fo1_amd import ops
    torch.manual_seed(31)
    H, KV, D, Lmax = 16, 2, 128, 512
    kc = torch.zeros(KV, Lmax, D, dtype=BF, device="cuda")
    vt = torch.zeros(KV * D, Lmax, dtype=BF, device="cuda")
    for n in (1, 63, 64, 65, 300, 512):
        k = (torch.randn(n, KV * D) * 1.2).to(BF).cuda()
        v = torch.randn(n, KV * D).to(BF).cuda()
        kc.zero_(); vt.zero_()
        kc[:, :n] = k.view(n, KV, D).permute(1, 0, 2)
        ops.transpose_into(v, vt, 0)
        q = (torch.randn(1, H * D) * 1.2).to(BF).cuda()
        kv_len = torch.tensor([n], dtype=torch.int32, device="cuda")
        got = ops.attention_decode(q, kc, vt, kv_len, Lmax, H, KV, D, 1 / math.sqrt(D))
        ref = attn_ref(q.float().cpu().reshape(1, H, D).expand(n, H, D).contiguous(), k.float().cpu().reshape(n, KV, D),
                       v.float().cpu().reshape(n, KV, D), [(0, n)], False, 1 / math.sqrt(D))[0].reshape(1, H * D)
        err = (got.float().cpu() - ref).abs().max()
        assert err < 3e-2, f"decode attention n={n}: max err {err:.4g}"
    big = torch.randn(151936).to(BF)
    big[150000] = big.max() + 2
    big[151000] = big[150000]
    assert ops.argmax(big.cuda()).item() == 150000, "two-stage argmax: first index among ties"


def test_gemv_fused_rmsnorm_and_ksplit(ab_library):
    """fo1_gemv_bf16 with the RMSNorm folded into its prologue == rmsnorm kernel + GEMV, bit for bit; deep-K shapes
    take the K-split-across-waves variant and must match the tile kernel."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(40)
    for (M, N, K, act) in [(1, 2560, 2048, 0), (1, 22016, 2048, 3), (1, 2048, 4096, 0), (3, 512, 4096, 1), (1, 151936, 2048, 0)]:
        x = (torch.randn(M, K) * 2).to(BF).cuda()
        w = (torch.randn(N, K) * 0.05).to(BF).cuda()
        nw = (1 + 0.1 * torch.randn(K)).to(BF).cuda()
        fused = ops.gemv(x, w, act=act, norm_weight=nw, norm_eps=1e-6)
        ref = ops.gemv(ops.rmsnorm(x, nw, 1e-6), w, act=act)
        assert torch.equal(fused, ref), f"fused norm differs M={M} N={N} K={K}"
        L.load().fo1_gemm_set_gemv(0)
        try:
            tile = ops.gemm(ops.rmsnorm(x, nw, 1e-6), w, act=act)
        finally:
            L.load().fo1_gemm_set_gemv(1)
        scale = tile.float().abs().max().item()
        assert (fused.float() - tile.float()).abs().max().item() <= 2 ** -7 * scale + 1e-3


def test_decode_qkv_post_matches_rope_and_transpose():
    from vlm_fo1_amd import ops
    torch.manual_seed(41)
    H, KV, HD, Lmax = 16, 2, 128, 128
    p = torch.arange(Lmax).view(1, -1).expand(3, -1)
    from vlm_fo1_amd.llm import mrope_tables
    cos, sin = mrope_tables(p, HD, 1e6, (16, 24, 24))
    cos, sin = cos.cuda(), sin.cuda()
    for pos, row in [(0, 0), (37, 21), (100, 90)]:
        qkv = torch.randn(1, (H + 2 * KV) * HD).to(BF).cuda()
        ref = qkv.clone()
        kc_ref = torch.zeros(KV, Lmax, HD, dtype=BF, device="cuda")
        vt_ref = torch.zeros(KV * HD, Lmax, dtype=BF, device="cuda")
        ops.rope_llm(ref, H + KV, HD, cos[row:row + 1].contiguous(), sin[row:row + 1].contiguous(), kcache=kc_ref, k_first_head=H, pos0=pos)
        ops.transpose_into(ref[:, (H + KV) * HD:], vt_ref, pos)
        kc = torch.zeros_like(kc_ref); vt = torch.zeros_like(vt_ref)
        st = torch.tensor([pos, row, 0, 0, pos, pos + 1, 0, pos + 1], dtype=torch.int32, device="cuda")
        ops.decode_qkv_post(qkv, H, KV, HD, cos, sin, st, kc, vt)
        assert torch.equal(qkv, ref) and torch.equal(kc, kc_ref) and torch.equal(vt, vt_ref)
        ops.decode_advance(st)
        assert st.tolist() == [pos + 1, row + 1, 0, 0, pos + 1, pos + 2, 0, pos + 2]


def test_qkv_post_fused_equals_separate_kernels():
    """The one-launch prefill post-processing (rope + K append + V^T) must be bit-identical to the separately
    tested rope_* and transpose kernels."""
    from vlm_fo1_amd import ops
    torch.manual_seed(21)
    # LLM: 16 q heads, 2 kv heads, hd 128, ragged L, append at pos0
    L, H, KV, HD, pos0, max_seq = 139, 16, 2, 128, 37, 256
    qkv = torch.randn(L, (H + 2 * KV) * HD).to(BF).cuda()
    cos = torch.randn(L, HD).to(BF).cuda()
    sin = torch.randn(L, HD).to(BF).cuda()
    a, b = qkv.clone(), qkv.clone()
    kc_a = torch.zeros(KV, max_seq, HD, dtype=BF, device="cuda"); kc_b = torch.zeros_like(kc_a)
    vt_a = torch.zeros(KV * HD, max_seq, dtype=BF, device="cuda"); vt_b = torch.zeros_like(vt_a)
    ops.rope_llm(a, H + KV, HD, cos, sin, kcache=kc_a, k_first_head=H, pos0=pos0)
    ops.transpose_into(a[:, (H + KV) * HD:], vt_a, col0=pos0)
    ops.qkv_post_llm(b, H, KV, HD, cos, sin, kc_b, vt_b, pos0)
    assert torch.equal(a, b) and torch.equal(kc_a, kc_b) and torch.equal(vt_a, vt_b)
    assert vt_b[:, :pos0].abs().sum() == 0 and vt_b[:, pos0 + L:].abs().sum() == 0
    # ViT: 16 heads, hd 80
    S, H, hd = 203, 16, 80
    d = H * hd
    qkv = torch.randn(S, 3 * d).to(BF).cuda()
    cos = torch.randn(S, hd // 2).cuda(); sin = torch.randn(S, hd // 2).cuda()
    a, b = qkv.clone(), qkv.clone()
    vt_a = torch.zeros(d, 256, dtype=BF, device="cuda"); vt_b = torch.zeros_like(vt_a)
    ops.rope_vit(a, H, hd, cos, sin)
    ops.transpose_into(a[:, 2 * d:], vt_a, 0)
    ops.qkv_post_vit(b, H, hd, cos, sin, vt_b)
    assert torch.equal(a, b) and torch.equal(vt_a, vt_b)


P8_SHAPES = [
    # (M, N, K, bias, res, act, splits)   — fo1 tile code 5: 256x256 ping-pong kernel (32x32x16 MFMA, counted-vmcnt LDS-DMA ring)
    (700, 520, 64, True, False, 0, 1),         # one k tile (prologue-only pipeline), ragged M and N
    (300, 256, 128, False, True, 0, 1),        # two k tiles
    (1025, 1028, 192, True, True, 0, 1),       # three k tiles (odd), one row / one column group past the tile edge
    (1564, 3840, 1280, True, False, 0, 1),     # ViT qkv, one image
    (5208, 2560, 2048, True, False, 0, 1),     # LLM qkv, 8 images x 651 rows
    (3128, 1280, 3456, True, True, 0, 1),      # ViT down + residual, 2 images
    (2604, 2048, 5888, True, False, 1, 1),     # GELU epilogue
    (777, 512, 320, True, True, 2, 1),         # SiLU epilogue + residual
    (2604, 2048, 11008, False, True, 0, 3),    # LLM down, split-K 3 (fp32 partials + fixed-order reduce)
    (1300, 1024, 1024, True, False, 0, 2),     # split-K 2
]


@pytest.mark.parametrize("sched", [0, 1, 3])
def test_gemm_p8_256x256(sched, ab_library):
    """sched bit 0: 0 = four phases per K tile, 1 = two fat phases with the LDS-DMA issued between MFMAs (gemm_bt_p4_kernel);
    bit 1 set = fragment-shaped epilogue stores instead of the LDS-staged coalesced epilogue."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(55)
    try:
        L.check(L.load().fo1_gemm_set_big_schedule(sched), "schedule")
        for (M, N, K, hb, hr, act, splits) in P8_SHAPES:
            L.check(L.load().fo1_gemm_set_variant(0, 5), "variant")
            L.check(L.load().fo1_gemm_set_splitk(splits), "splitk")
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            res = torch.randn(M, N).to(BF).cuda() if hr else None
            got = ops.gemm(a, w, bias, res, act)
            ref = gemm_ref(a, w, bias, res, act)
            scale = ref.abs().max().item()
            err = (got.float().cpu() - ref).abs().max().item()
            rel = (got.float().cpu() - ref).norm() / ref.norm()
            assert err <= 2e-2 * scale + 1e-3 and rel < 4e-3, f"p8 gemm {M}x{N}x{K} act={act} splits={splits}: max err {err:.4g} (scale {scale:.4g}), rel fro {rel:.4g}"
            # race screen: the pipeline's LDS hazards (DMA landing vs ds_read, restaging vs the lagging half's reads) would show up as
            # run-to-run differences; 12 more launches must be bit-identical
            for _ in range(12):
                again = ops.gemm(a, w, bias, res, act)
                assert torch.equal(again, got), f"p8 gemm {M}x{N}x{K}: two launches differ (race)"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_big_schedule(1)


@pytest.mark.parametrize("sched", [0, 1, 3])
def test_gemm_p8_swiglu_and_one_hot(sched, ab_library):
    """(a) the interleaved-SwiGLU epilogue on 32x32 fragments against the unfused reference; (b) an A = one-hot-rows GEMM whose exact
    answer is a row of W: catches any row/column or k-chunk mix-up exactly (no tolerance)."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(56)
    try:
        L.check(L.load().fo1_gemm_set_big_schedule(sched), "schedule")
        L.check(L.load().fo1_gemm_set_variant(0, 5), "variant")
        M, K, Fh = 1564, 1280, 3456
        a = (torch.randn(M, K) * 0.5).to(BF).cuda()
        wg = (torch.randn(Fh, K) * 0.05).to(BF)
        wu = (torch.randn(Fh, K) * 0.05).to(BF)
        bg, bu = (torch.randn(Fh) * 0.1).to(BF), (torch.randn(Fh) * 0.1).to(BF)
        w = ops.interleave_gate_up(wg, wu).cuda()
        b = ops.interleave_gate_up(bg, bu).cuda()
        got = ops.gemm(a, w, b, act=ops.ACT_SWIGLU16)
        g = rb(a.float().cpu() @ wg.float().t() + bg.float())
        u = rb(a.float().cpu() @ wu.float().t() + bu.float())
        ref = rb(rb(torch.nn.functional.silu(g)) * u)
        err = (got.float().cpu() - ref).abs().max().item()
        assert got.shape == (M, Fh) and err <= 2e-2 * ref.abs().max().item() + 1e-3, f"p8 swiglu: max err {err:.4g}"
        # one-hot A: row m picks column (7 m + 3) mod K of W  ->  C[m, n] = W[n, k(m)] exactly
        M, N, K = 1100, 768, 448
        kk = (torch.arange(M) * 7 + 3) % K
        a = torch.zeros(M, K)
        a[torch.arange(M), kk] = 1.0
        w = (torch.randn(N, K)).to(BF)
        got = ops.gemm(a.to(BF).cuda(), w.cuda())
        assert torch.equal(got.cpu(), w[:, kk].t().contiguous()), "p8 gemm: one-hot A does not select W columns exactly"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_big_schedule(1)


def test_gemm_persistent_tile_loop_equals_one_tile_per_workgroup(ab_library):
    """gemm_bt_p4p_kernel (256 persistent workgroups, the K loop running on across output tiles, epilogue through 4 KiB LDS strips)
    against gemm_bt_p4_kernel (one tile per workgroup): the per-tile arithmetic is the same instruction sequence, so the results
    must be BIT-IDENTICAL — for every epilogue, with ragged M / N edges, for more than 256 tiles (the persistent form's trigger) —
    and identical run to run (race screen for the hand-over of the LDS image between tiles); plus the usual check against fp32."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(57)
    shapes = [  # M, N, K, bias, residual, act
        (5216, 22016, 2048, False, False, ops.ACT_SWIGLU16),     # LLM gate/up of an 8-image pass: 1806 tiles, 7 per workgroup
        (12512, 3840, 1280, True, False, ops.ACT_NONE),          # ViT qkv: 735 tiles
        (18768, 1280, 3456, True, True, ops.ACT_NONE),           # ViT down + residual (12 images): 370 tiles
        (9600, 4096, 1024, True, False, ops.ACT_GELU),           # DaViT fc1 + GELU: 608 tiles
        (6000, 3000, 1280, True, True, ops.ACT_SILU),            # ragged M and N edges: 24 x 12 tiles
        (7000, 2816, 128, False, True, ops.ACT_NONE),            # two k-tiles only: every k-tile is a hand-over tile
    ]
    try:
        for (M, N, K, hb, hr, act) in shapes:
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            res = torch.randn(M, N).to(BF).cuda() if hr else None
            L.check(L.load().fo1_gemm_set_variant(0, 5), "variant")
            L.check(L.load().fo1_gemm_set_big_schedule(1), "schedule")          # one tile per workgroup (default)
            one = ops.gemm(a, w, bias, res, act)
            L.check(L.load().fo1_gemm_set_big_schedule(1 | 4), "schedule")      # persistent tile loop
            per = ops.gemm(a, w, bias, res, act)
            assert torch.equal(per, one), f"persistent != one-tile-per-workgroup at {M}x{N}x{K} act={act}: {(per.float() - one.float()).abs().max().item():.4g}"
            for _ in range(8):
                assert torch.equal(ops.gemm(a, w, bias, res, act), per), f"persistent gemm {M}x{N}x{K}: two launches differ (race)"
            if act != ops.ACT_SWIGLU16:
                ref = gemm_ref(a, w, bias, res, act)
                err = (per.float().cpu() - ref).abs().max().item()
                assert err <= 2e-2 * ref.abs().max().item() + 1e-3, f"persistent gemm {M}x{N}x{K}: max err {err:.4g}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_big_schedule(1)


def test_gemm_coalesced_epilogue_equals_fragment_epilogue_bitwise(ab_library):
    """The LDS-staged epilogue of the 256x256 kernel (round 3: bias pieces and all 16 residual rows loaded up front from clamped addresses,
    pairwise rounding, straight write-out sweeps) against the fragment-shaped epilogue32 (each load next to its use, element-wise
    rounding): the same rounding points, so the outputs must be bit-identical — ragged M / N edges (clamped rows / columns), every
    activation, with and without bias / residual, and the interleaved SwiGLU form."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(61)
    shapes = [(700, 520, 64, True, True, 0), (1025, 1032, 192, True, True, 0), (300, 256, 128, False, True, 0), (1564, 3840, 1280, True, False, 0),
              (777, 512, 320, True, True, 2), (2604, 2048, 1024, True, False, 1), (519, 776, 256, False, False, 1), (1300, 1288, 512, False, True, 2)]
    try:
        L.check(L.load().fo1_gemm_set_variant(0, 5), "variant")
        for (M, N, K, hb, hr, act) in shapes:
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
            res = torch.randn(M, N).to(BF).cuda() if hr else None
            L.check(L.load().fo1_gemm_set_big_schedule(1), "schedule")
            coal = ops.gemm(a, w, bias, res, act)
            L.check(L.load().fo1_gemm_set_big_schedule(3), "schedule")
            frag = ops.gemm(a, w, bias, res, act)
            assert torch.equal(coal, frag), f"{M}x{N}x{K} act={act}: coalesced != fragment epilogue, max diff {(coal.float() - frag.float()).abs().max().item():.4g}"
        M, K, Fh = 1301, 512, 1312
        a = (torch.randn(M, K) * 0.5).to(BF).cuda()
        w = ops.interleave_gate_up((torch.randn(Fh, K) * 0.05).to(BF), (torch.randn(Fh, K) * 0.05).to(BF)).cuda()
        for b in (None, ops.interleave_gate_up((torch.randn(Fh) * 0.1).to(BF), (torch.randn(Fh) * 0.1).to(BF)).cuda()):
            L.check(L.load().fo1_gemm_set_big_schedule(1), "schedule")
            coal = ops.gemm(a, w, b, act=ops.ACT_SWIGLU16)
            L.check(L.load().fo1_gemm_set_big_schedule(3), "schedule")
            frag = ops.gemm(a, w, b, act=ops.ACT_SWIGLU16)
            assert torch.equal(coal, frag), f"swiglu bias={b is not None}: coalesced != fragment epilogue"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_big_schedule(1)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_gemm_small_tile_vector_epilogue_equals_general_epilogue_bitwise(tile, ab_library):
    """epilogue_vec (16x16-fragment kernels: every bias / residual piece loaded up front, activation as a template parameter) is taken
    when bias, residual and C are 8-byte aligned; a bias or residual view that starts 2 bytes off the alignment goes through the general
    epilogue (loads next to their use).  Same operands, same rounding points: bit-identical outputs."""
    from vlm_fo1_amd import lib as L, ops
    torch.manual_seed(62 + tile)
    shapes = [(77, 132, 128, True, True, 2), (300, 516, 256, True, True, 0), (651, 2048, 192, True, False, 1), (130, 1280, 320, False, True, 0),
              (1564, 1284, 128, True, True, ops.ACT_RELU)]
    try:
        L.check(L.load().fo1_gemm_set_variant(2, tile), "variant")
        for (M, N, K, hb, hr, act) in shapes:
            a = (torch.randn(M, K) * 0.5).to(BF).cuda()
            w = (torch.randn(N, K) * 0.05).to(BF).cuda()
            bias_v = (torch.randn(N) * 0.1).to(BF).cuda()
            res_v = torch.randn(M, N).to(BF).cuda()
            bias_store, res_store = torch.empty(N + 8, dtype=BF, device="cuda"), torch.empty(M * N + 8, dtype=BF, device="cuda")
            bias_store[1:N + 1].copy_(bias_v)
            res_store[1:M * N + 1].copy_(res_v.reshape(-1))
            outs = []
            for mis in (False, True):       # views that start 2 bytes off the 8-byte alignment take the general epilogue
                bias = (bias_store[1:N + 1] if mis else bias_v) if hb else None
                res = (res_store[1:M * N + 1].view(M, N) if mis else res_v) if hr else None
                outs.append(ops.gemm(a, w, bias, res, act))
            if len(outs) == 2:
                assert torch.equal(outs[0], outs[1]), f"tile {tile} {M}x{N}x{K} act={act}: vector != general epilogue, max diff {(outs[0].float() - outs[1].float()).abs().max().item():.4g}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)


@pytest.mark.parametrize("M,pos0", [(700, 16), (1290, 0)])
def test_qkv_proj_rope_llm_equals_gemm_plus_qkv_post(ab_library, M, pos0):
    """fo1_qkv_proj_rope_bf16 mode 0 (q/k/v projection with mRoPE + K-cache append + V^T write in the 256 x 256 GEMM's epilogue) against the two
    launches it replaces on the SAME GEMM kernel (tile pinned to 256 x 256): rotated q rows, the K cache and the V^T cache bit for bit — M not a
    multiple of 8 (row tails of the transposed 16-byte stores), a non-zero cache position."""
    from vlm_fo1_amd import lib as L, ops
    H, KV, D, K, cap = 16, 2, 128, 256, 2048
    g = torch.Generator().manual_seed(31 + M)
    x = (torch.randn(M, K, generator=g)).to(BF).cuda()
    w = (torch.randn((H + 2 * KV) * D, K, generator=g) * 0.08).to(BF).cuda()
    b = (torch.randn((H + 2 * KV) * D, generator=g) * 0.3).to(BF).cuda()
    ang = torch.rand(M, D, generator=g) * 6.28
    cos, sin = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    L.load().fo1_gemm_set_variant(0, 5)
    try:
        assert L.load().fo1_gemm_takes_big_tile(M, (H + 2 * KV) * D, K) == 1
        qkv = ops.gemm(x, w, b)
        kc_ref = torch.zeros(KV, cap, D, dtype=BF, device="cuda"); vt_ref = torch.zeros(KV * D, cap, dtype=BF, device="cuda")
        ops.qkv_post_llm(qkv, H, KV, D, cos, sin, kc_ref, vt_ref, pos0)
        kc = torch.zeros_like(kc_ref); vt = torch.zeros_like(vt_ref)
        out = ops.qkv_proj_rope(x, w, b, 0, H, KV, cos, sin, kc, pos0, vt)
        torch.cuda.synchronize()
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
    assert torch.equal(out[:, :H * D], qkv[:, :H * D]), f"rotated q: max diff {(out[:, :H * D].float() - qkv[:, :H * D].float()).abs().max().item():.4g}"
    assert torch.equal(kc, kc_ref), f"K cache: max diff {(kc.float() - kc_ref.float()).abs().max().item():.4g}"
    assert torch.equal(vt, vt_ref), f"V^T cache: max diff {(vt.float() - vt_ref.float()).abs().max().item():.4g}"
    assert kc_ref[:, pos0:pos0 + M].abs().sum() > 0 and vt_ref[:, pos0:pos0 + M].abs().sum() > 0


@pytest.mark.parametrize("M", [1001, 1536])
def test_qkv_proj_rope_vit_equals_gemm_plus_qkv_post(ab_library, M):
    """mode 1: the ViT's q/k/v projection on HEAD-MAJOR weight rows (per head [q 80 | k 80 | v 80 | 16 zero rows] = one 256-column tile, so that
    a head's rotate-half pairs never straddle two workgroups) with the fp32 2-D RoPE and the V -> V^T copy in the epilogue, against fo1_gemm_bf16 +
    fo1_qkv_post_vit_bf16 on the reference layout [q heads | k heads | v heads]: the same numbers, bit for bit, at their new addresses."""
    from vlm_fo1_amd import lib as L, ops
    H, D, K = 16, 80, 256
    d = H * D
    g = torch.Generator().manual_seed(41 + M)
    x = (torch.randn(M, K, generator=g)).to(BF).cuda()
    w = (torch.randn(3 * d, K, generator=g) * 0.08).to(BF)
    b = (torch.randn(3 * d, generator=g) * 0.3).to(BF)
    ang = torch.rand(M, D // 2, generator=g) * 6.28
    cos, sin = ang.cos().cuda(), ang.sin().cuda()
    Sp = (M + 63) // 64 * 64
    L.load().fo1_gemm_set_variant(0, 5)
    try:
        qkv = ops.gemm(x, w.cuda(), b.cuda())
        vt_ref = torch.zeros(d, Sp, dtype=BF, device="cuda")
        ops.qkv_post_vit(qkv, H, D, cos, sin, vt_ref)
        vt = torch.zeros_like(vt_ref)
        out = ops.qkv_proj_rope(x, ops.head_major_qkv(w, H, D).cuda(), ops.head_major_qkv(b, H, D).cuda(), 1, H, H, cos, sin, None, 0, vt)
        torch.cuda.synchronize()
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
    o = out.view(M, H, 256)
    assert torch.equal(o[:, :, :D].reshape(M, d), qkv[:, :d]), "rotated q"
    assert torch.equal(o[:, :, D:2 * D].reshape(M, d), qkv[:, d:2 * d]), "rotated k"
    assert torch.equal(vt, vt_ref), f"V^T: max diff {(vt.float() - vt_ref.float()).abs().max().item():.4g}"
    # and the attention reads the head-major layout through its head stride: same output as on the reference layout
    items = ops.make_items([(0, M)], "cuda", block=64)
    a_ref = ops.attention(qkv[:, :d], qkv[:, d:2 * d], vt_ref, items, H, H, D, D ** -0.5, False)
    a_hm = ops.attention(out, out[:, D:], vt, items, H, H, D, D ** -0.5, False, qk_head_stride=256)
    assert torch.equal(a_ref, a_hm)
    items = ops.make_items([(0, M)], "cuda", block=256)
    assert torch.equal(ops.attention(qkv[:, :d], qkv[:, d:2 * d], vt_ref, items, H, H, D, D ** -0.5, False),
                       ops.attention(out, out[:, D:], vt, items, H, H, D, D ** -0.5, False, qk_head_stride=256))


@pytest.mark.parametrize("cfg", [
    dict(name="fpn_s1", B=2, H=40, W=52, Cin=512, Cout=512, stride=1, act=0, bias=False),
    dict(name="davit_s2", B=3, H=61, W=83, Cin=256, Cout=512, stride=2, act=0, bias=True),
    dict(name="gelu_1024", B=1, H=48, W=48, Cin=1024, Cout=256, stride=2, act=1, bias=True),
])
def test_conv3x3_implicit_gemm_equals_im2col_gemm(ab_library, cfg):
    """fo1_layernorm_rows_bf16 + fo1_conv3x3_gemm_bf16 (LayerNorm written straight into the zero-padded map, 3x3 convolution as an implicit GEMM
    on the 256 x 256 kernel: the nine taps are nine K-tile address offsets, no column matrix) against layernorm + fo1_im2col_bf16 + fo1_gemm_bf16
    pinned to the same kernel: bit for bit — odd sizes (stride-2 borders), several images (no leak across image borders), bias / GELU."""
    from vlm_fo1_amd import lib as L, ops
    B, H, W, Cin, Cout, s = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"], cfg["stride"]
    g = torch.Generator().manual_seed(51)
    x = torch.randn(B * H * W, Cin, generator=g).to(BF).cuda()
    nw, nb = (1 + 0.1 * torch.randn(Cin, generator=g)).to(BF).cuda(), (0.1 * torch.randn(Cin, generator=g)).to(BF).cuda()
    w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.02).to(BF).cuda()
    b = (torch.randn(Cout, generator=g) * 0.2).to(BF).cuda() if cfg["bias"] else None
    L.load().fo1_gemm_set_variant(0, 5)
    try:
        y = ops.layernorm(x, nw, nb, 1e-6)
        col, Ho, Wo = ops.im2col(y, H, W, 3, 3, s, 1, batch=B)
        ref = ops.gemm(col, w, b, act=cfg["act"])
        pl = ops.conv3x3_plan(((H, W),) * B, s, Cin, "cuda")
        assert pl.out_hw == [(Ho, Wo)] * B and pl.M_out == B * Ho * Wo
        yp = ops.layernorm_rows(x, nw, nb, 1e-6, torch.zeros(pl.pad_rows, Cin, dtype=BF, device="cuda"), pl.rowmap)
        got = ops.conv3x3_gemm(yp, pl, w, b, act=cfg["act"])
        torch.cuda.synchronize()
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
    assert torch.equal(yp[pl.rowmap.long()], y), "layernorm_rows: the interior of the padded map"
    assert got.shape == ref.shape and torch.equal(got, ref), f"{cfg['name']}: max diff {(got.float() - ref.float()).abs().max().item():.4g}"


def test_mfma_clock_probe_reports_a_plausible_clock(ab_library):
    """fo1_mfma_clock_probe (csrc/probe.hip; an instrument of include/fo1_ab.h since round 5 — bench.py loads the test / bench build for it after the timed region): cycles / wall ticks of a register-resident MFMA loop = a clock inside the part's
    DVFS range, 32 cycles per 32x32x16 bf16 MFMA per SIMD (two waves share one), and zero operands never clock lower than random ones."""
    from vlm_fo1_amd import ops
    rnd = ops.mfma_clock_probe(1, iters=400)
    zero = ops.mfma_clock_probe(0, iters=400)
    for r in (rnd, zero):
        assert 0.8 <= r["clock_ghz"] <= 2.6, r
        cycles_per_mfma = r["clock_ghz"] * 1e3 * r["us"] / (400 * 32 * 2)
        assert 31.0 <= cycles_per_mfma <= 36.0, (r, cycles_per_mfma)
    assert zero["tflops"] >= 0.97 * rnd["tflops"]


def test_pipelined_single_tile_attention_equals_the_general_kernel_bitwise():
    """Round 6: fo1_attention_windows_bf16 (a workgroup walks 4 single-tile items with the next item's loads in flight) == fo1_attention_bf16 with
    q_block 64 on the same list, bit for bit: window lengths 4..64 (ragged edge windows), item counts that are not a multiple of 4, packed and
    head-major (stride 256) q/k layouts."""
    from vlm_fo1_amd import ops
    torch.manual_seed(41)
    H, hd = 16, 80
    d = H * hd
    for lens in ([64, 64, 32, 16, 4, 60, 64, 8, 48, 36, 64], [64] * 9, [12]):
        S = sum(lens)
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + n)
        segs = list(zip(cu[:-1], cu[1:]))
        assert ops.single_tile_items(segs, hd)
        items = ops.make_items(segs, "cuda", block=64)
        items.single_tile = True
        Sp = (S + 63) // 64 * 64
        for head_major in (False, True):
            qkv = (torch.randn(S, 3 * d) * 1.2).to(torch.bfloat16).cuda()
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            vt = torch.zeros(d, Sp, dtype=torch.bfloat16, device="cuda")
            vt[:, :S] = v.t()
            if head_major:
                hm = torch.zeros(S, H * 256, dtype=torch.bfloat16, device="cuda")
                hm.view(S, H, 256)[:, :, :hd] = q.view(S, H, hd)
                hm.view(S, H, 256)[:, :, hd:2 * hd] = k.view(S, H, hd)
                qa, ka, hs = hm, hm[:, hd:], 256
            else:
                qa, ka, hs = q, k, None
            got = ops.attention_windows(qa, ka, vt, items, H, hd, hd ** -0.5, qk_head_stride=hs)
            ref = ops.attention(qa, ka, vt, items, H, H, hd, hd ** -0.5, False, qk_head_stride=hs)
            assert torch.equal(got, ref), f"{len(lens)} windows, head_major={head_major}: {int((got != ref).sum())} elements differ"
