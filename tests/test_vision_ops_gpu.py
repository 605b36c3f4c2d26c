"""GPU parity of the DaViT / SimpleFPN / splice data-movement kernels vs plain torch on the CPU.
Pure copies must be bit-exact; arithmetic ops follow the 1-bf16-ulp rule of test_ops_gpu.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import close_bf16, rb

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def tm(x_nchw):  # [1,C,H,W] -> token-major [H*W, C]
    return x_nchw[0].permute(1, 2, 0).reshape(-1, x_nchw.shape[1]).contiguous()


def test_dwconv3x3_residual():
    from vlm_fo1_amd import ops
    torch.manual_seed(0)
    for (H, W, C) in [(13, 16, 256), (25, 32, 64), (5, 3, 8)]:
        x = torch.randn(1, C, H, W).to(BF)
        w = (torch.randn(C, 1, 3, 3) * 0.3).to(BF)
        b = (torch.randn(C) * 0.1).to(BF)
        ref = x.float() + rb(F.conv2d(x.float(), w.float(), b.float(), padding=1, groups=C))
        w9c = w.reshape(C, 9).t().contiguous()
        got = ops.dwconv3x3_res(tm(x).cuda(), w9c.cuda(), b.cuda(), H, W)
        close_bf16(got, rb(tm(ref)), ulps=1.01, atol=1e-5, what=f"dwconv {H}x{W}x{C}")


@pytest.mark.parametrize("H,W,C,K,s,p", [(20, 27, 8, 7, 4, 3), (25, 32, 64, 3, 2, 1), (12, 12, 128, 3, 1, 1)])
def test_im2col_conv_equivalence(H, W, C, K, s, p):
    """im2col + GEMM == conv2d with the weight re-laid out [Cout][ky][kx][Cin]."""
    from vlm_fo1_amd import ops
    torch.manual_seed(1)
    Co = 64
    x = torch.randn(1, C, H, W).to(BF)
    w = (torch.randn(Co, C, K, K) * 0.05).to(BF)
    b = (torch.randn(Co) * 0.1).to(BF)
    ref = rb(F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=p))
    col, Ho, Wo = ops.im2col(tm(x).cuda(), H, W, K, K, s, p)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    unf = F.unfold(x.float(), K, padding=p, stride=s)[0].t().reshape(Ho * Wo, C, K * K).permute(0, 2, 1).reshape(Ho * Wo, -1)
    assert torch.equal(col.float().cpu(), unf), "im2col is a pure copy: must be bit exact"
    wg = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
    got = ops.gemm(col, wg.cuda(), b.cuda())
    err = (got.float().cpu() - tm(ref)).abs().max()
    assert err < 2e-2 * tm(ref).abs().max() + 1e-3, f"conv via im2col+gemm: max err {err:.4g}"


def test_window_partition_reverse():
    from vlm_fo1_amd import ops
    torch.manual_seed(2)
    for (H, W, C, ws) in [(25, 32, 64, 12), (24, 36, 32, 12), (5, 7, 8, 12)]:
        x = torch.randn(H, W, C).to(BF)
        pad_b, pad_r = (ws - H % ws) % ws, (ws - W % ws) % ws
        xp = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = xp.shape[:2]
        ref = xp.view(Hp // ws, ws, Wp // ws, ws, C).permute(0, 2, 1, 3, 4).reshape(-1, C)
        got = ops.window_partition(x.reshape(-1, C).cuda(), H, W, ws)
        assert torch.equal(got.cpu(), ref)
        yw = torch.randn_like(ref)
        y = yw.view(Hp // ws, Wp // ws, ws, ws, C).permute(0, 2, 1, 3, 4).reshape(Hp, Wp, C)[:H, :W]
        refo = rb(x.float() + y.float())
        goto = ops.window_reverse_add(yw.cuda(), x.reshape(-1, C).cuda(), H, W, ws)
        assert torch.equal(goto.float().cpu(), refo.reshape(-1, C))


def test_channel_attention():
    from vlm_fo1_amd import ops
    torch.manual_seed(3)
    for (N, C) in [(300, 64), (1000, 256), (37, 32)]:
        G = C // 32
        qkv = torch.randn(N, 3 * C).to(BF)
        x = qkv.float().reshape(1, N, 3, G, 32).permute(2, 0, 3, 1, 4)
        q, k, v = x[0], x[1], x[2]
        att = rb(rb(rb(q * (float(N) ** -0.5)).transpose(-1, -2) @ k).softmax(-1))
        ref = rb((att @ v.transpose(-1, -2)).transpose(-1, -2)).transpose(1, 2).reshape(N, C)
        got = ops.channel_attention(qkv.cuda(), C)
        err = (got.float().cpu() - ref).abs().max()
        assert err < 0.03 * ref.abs().max() + 2e-3, f"channel attention N={N} C={C}: max err {err:.4g} (scale {ref.abs().max():.3g})"


def test_channel_attention_matrix_core_kernels_against_the_fp32_fma_kernels():
    """Round 6: the Gram matrices and the attention product run on v_mfma_f32_32x32x16_bf16.  Products of bf16 values are exact either way;
    the two forms differ in the order of the fp32 additions only, so after the bf16 roundings (softmax input and output, result) almost every
    output element is equal and none is more than a few bf16 steps away.  DaViT stage shapes incl. ragged token counts and stacked images."""
    from vlm_fo1_amd import lib as _L, ops
    torch.manual_seed(33)
    for (B, N, C) in [(1, 1200, 1024), (3, 300, 256), (2, 37, 64), (1, 513, 32), (2, 4800, 512)]:
        qkv = torch.randn(B * N, 3 * C).to(BF).cuda()
        new = ops.channel_attention(qkv, C, batch=B)
        with _L.use_ab():
            _L.load().fo1_channel_attention_set_impl(0)
            try:
                old = ops.channel_attention(qkv, C, batch=B)
            finally:
                _L.load().fo1_channel_attention_set_impl(1)
        d = (new.float() - old.float()).abs()
        frac_equal = float((d == 0).float().mean())
        assert frac_equal > 0.97, f"B={B} N={N} C={C}: only {frac_equal:.4f} of the elements equal"
        assert float(d.max()) <= 0.02 * float(old.float().abs().max()) + 1e-3, f"B={B} N={N} C={C}: max diff {float(d.max()):.4g}"
        # run to run: bitwise
        assert torch.equal(new, ops.channel_attention(qkv, C, batch=B))


def test_window_attention_on_the_qkv_rows_against_a_torch_reference_and_the_general_kernel():
    """Round 6: fo1_window_attention_bf16 — DaViT's WindowAttention core (modeling_davit.py:225-282) for head dim 32 on the q/k/v GEMM's
    [windows * tokens, 3C] rows.  Against an fp32 torch softmax(q k^T / sqrt(32)) v per window and head (bf16 tolerance), against the general
    attention kernel it replaces (same rounding points: within a couple of bf16 steps), window counts / token counts that are not multiples of
    the tile sizes, and run to run bitwise."""
    from vlm_fo1_amd import ops
    torch.manual_seed(34)
    for (n_win, wtok, heads) in [(7, 144, 8), (3, 144, 32), (5, 64, 4), (2, 37, 12), (1, 160, 16), (4, 1, 8)]:
        C = heads * 32
        n = n_win * wtok
        qkv = (torch.randn(n, 3 * C) * 1.5).to(BF).cuda()
        got = ops.window_attention(qkv, C, heads, wtok, 32 ** -0.5)
        x = qkv.float().view(n_win, wtok, 3, heads, 32).permute(2, 0, 3, 1, 4)          # [3][win][head][tok][32]
        att = torch.softmax(x[0] @ x[1].transpose(-1, -2) * 32 ** -0.5, dim=-1)
        ref = (att @ x[2]).permute(0, 2, 1, 3).reshape(n, C)
        err = (got.float() - ref).abs().max()
        assert err < 0.02 * ref.abs().max() + 2e-3, f"{n_win} x {wtok} x {heads}: max err {float(err):.4g}"
        assert torch.equal(got, ops.window_attention(qkv, C, heads, wtok, 32 ** -0.5))
        # the general kernel over an item list with a transposed V
        n_pad = (n + 63) // 64 * 64
        vt = torch.zeros(C, n_pad, dtype=BF, device="cuda")
        ops.transpose_into(qkv[:, 2 * C:], vt, 0)
        segs = [(i * wtok, (i + 1) * wtok) for i in range(n_win)]
        items = ops.make_items(segs, "cuda", block=ops.pick_q_block(segs, heads))
        old = ops.attention(qkv[:, :C], qkv[:, C:2 * C], vt, items, heads, heads, 32, 32 ** -0.5, False)
        d = (got.float() - old.float()).abs()
        assert float(d.max()) <= 0.02 * float(old.float().abs().max()) + 1e-3, f"{n_win} x {wtok} x {heads}: vs the general kernel {float(d.max()):.4g}"


def test_window_attention_on_unpartitioned_rows_equals_partition_attention_reverse_bitwise():
    """Round 6: fo1_window_attention_map_bf16 finds a window's tokens among the PIXEL rows (no window_partition, no padded rows in the GEMMs, no
    window_reverse) and reads the q/k/v bias row for the reference's zero-padded tokens.  The whole spatial-attention sub-block — q/k/v GEMM,
    attention, proj GEMM + residual — must equal window_partition -> GEMM -> fo1_window_attention_bf16 -> GEMM -> window_reverse_add bit for bit:
    sizes that are not multiples of the window, several images, and the ragged entry against per-image calls."""
    from vlm_fo1_amd import lib as _L, ops
    torch.manual_seed(36)
    ws = 12
    with _L.use_ab():
        _L.load().fo1_gemm_set_variant(2, 1); _L.load().fo1_gemm_set_splitk(1)     # one tile shape for every row count (the parity tests' determinism pin)
        try:
            for (B, H, W, heads) in [(2, 40, 30, 8), (1, 20, 15, 16), (3, 13, 25, 4), (1, 12, 12, 8), (2, 5, 7, 4)]:
                C = heads * 32
                n = B * H * W
                h = torch.randn(n, C).to(BF).cuda()
                x = torch.randn(n, C).to(BF).cuda()
                wq, bq = (torch.randn(3 * C, C) * 0.08).to(BF).cuda(), (torch.randn(3 * C) * 0.3).to(BF).cuda()
                wp, bp = (torch.randn(C, C) * 0.08).to(BF).cuda(), (torch.randn(C) * 0.1).to(BF).cuda()
                # partition form
                hw = ops.window_partition(h, H, W, ws, batch=B)
                att_w = ops.window_attention(ops.gemm(hw, wq, bq), C, heads, ws * ws, 32 ** -0.5)
                ref = ops.window_reverse_add(ops.gemm(att_w, wp, bp), x, H, W, ws, batch=B)
                # map form
                qkv = ops.gemm(h, wq, bq)
                att = ops.window_attention_map(qkv, C, heads, ws, H, W, B, bq, 32 ** -0.5)
                got = ops.gemm(att, wp, bp, residual=x)
                assert torch.equal(got, ref), f"{B} x {H}x{W} x {heads} heads: {int((got != ref).sum())} elements differ"
            # ragged: three images of different sizes in one call == the per-image calls
            heads, C = 8, 256
            sizes = [(40, 30), (13, 25), (7, 5)]
            qs = [(torch.randn(a * b, 3 * C) * 0.7).to(BF).cuda() for a, b in sizes]
            bq = (torch.randn(3 * C) * 0.3).to(BF).cuda()
            rows, r0, w0 = [], 0, 0
            for a, b in sizes:
                nwy, nwx = -(-a // ws), -(-b // ws)
                rows.append((r0, a, b, w0, nwy, nwx))
                r0 += a * b
                w0 += nwy * nwx * ws * ws
            sg = ops.ImgSegs(rows, "cuda", max(a * b for a, b in sizes), r0, max(r[4] * r[5] * ws * ws for r in rows), w0)
            got = ops.window_attention_map_var(torch.cat(qs, 0), C, heads, ws, sg, bq, 32 ** -0.5)
            r = 0
            for (a, b), q in zip(sizes, qs):
                one = ops.window_attention_map(q, C, heads, ws, a, b, 1, bq, 32 ** -0.5)
                assert torch.equal(got[r:r + a * b], one), f"ragged image {a}x{b}"
                r += a * b
        finally:
            _L.load().fo1_gemm_set_variant(0, 0); _L.load().fo1_gemm_set_splitk(0)


def test_pixel_shuffle_maxpool_nchw_gather():
    from vlm_fo1_amd import ops
    torch.manual_seed(4)
    H, W, Co, Ci = 6, 9, 16, 24
    x = torch.randn(1, Ci, H, W).to(BF)
    wt = (torch.randn(Ci, Co, 2, 2) * 0.1).to(BF)
    b = (torch.randn(Co) * 0.1).to(BF)
    ref = rb(F.conv_transpose2d(x.float(), wt.float(), b.float(), stride=2))
    wg = wt.permute(2, 3, 1, 0).reshape(4 * Co, Ci).contiguous()  # rows (dy, dx, co)
    Cip = 64  # K must be a multiple of 8; pad to a friendly size
    a = torch.zeros(H * W, Cip, dtype=BF); a[:, :Ci] = tm(x)
    wgp = torch.zeros(4 * Co, Cip, dtype=BF); wgp[:, :Ci] = wg
    y4 = ops.gemm(a.cuda(), wgp.cuda(), b.repeat(4).cuda())
    got = ops.pixel_shuffle2(y4, H, W, Co)
    err = (got.float().cpu() - tm(ref)).abs().max()
    assert err < 2e-2, f"convT via gemm+pixel_shuffle: {err:.4g}"
    # maxpool
    xm = torch.randn(1, 32, 7, 10).to(BF)
    refm = F.max_pool2d(xm.float(), 2, 2)
    gotm = ops.maxpool2(tm(xm).cuda(), 7, 10)
    assert torch.equal(gotm.float().cpu(), tm(refm))
    # nchw -> hwc8
    img = torch.randn(3, 11, 13)
    g8 = ops.nchw_to_hwc8(img.cuda())
    assert torch.equal(g8[:, :3].cpu(), img.to(BF).permute(1, 2, 0).reshape(-1, 3)) and g8[:, 3:].abs().sum() == 0
    g8b = ops.nchw_to_hwc8(img.to(BF).cuda())
    assert torch.equal(g8b.cpu(), g8.cpu())
    # gather rows
    t0, t1, t2 = torch.randn(50, 64).to(BF), torch.randn(7, 64).to(BF), torch.randn(3, 64).to(BF)
    plan = torch.tensor([[0, 49], [1, 0], [1, 6], [2, 2], [0, 0], [2, 0]], dtype=torch.int32)
    gotg = ops.gather_rows(plan.cuda(), 64, t0.cuda(), t1.cuda(), t2.cuda())
    refg = torch.stack([t0[49], t1[0], t1[6], t2[2], t0[0], t2[0]])
    assert torch.equal(gotg.cpu(), refg)


def test_dwconv_layernorm_fused_equals_separate_kernels():
    """fo1_dwconv3x3_ln_bf16 == fo1_dwconv3x3_bf16 then fo1_layernorm_bf16, bit for bit (DaViT stage widths, ragged sizes)."""
    from vlm_fo1_amd import ops
    torch.manual_seed(31)
    for (H, W, C) in [(13, 17, 256), (9, 11, 512), (7, 5, 1024), (4, 5, 2048), (1, 1, 256), (3, 130, 512)]:
        x = torch.randn(H * W, C).to(torch.bfloat16).cuda()
        w9 = (torch.randn(9, C) * 0.2).to(torch.bfloat16).cuda()
        b = (torch.randn(C) * 0.1).to(torch.bfloat16).cuda()
        lw = (1 + 0.1 * torch.randn(C)).to(torch.bfloat16).cuda()
        lb = (0.1 * torch.randn(C)).to(torch.bfloat16).cuda()
        y_ref = ops.dwconv3x3_res(x, w9, b, H, W)
        h_ref = ops.layernorm(y_ref, lw, lb, 1e-5)
        y, h = ops.dwconv3x3_res_ln(x, w9, b, H, W, lw, lb, 1e-5)
        assert torch.equal(y, y_ref), f"{H}x{W}x{C}: conv output differs"
        assert torch.equal(h, h_ref), f"{H}x{W}x{C}: LayerNorm output differs ({int((h != h_ref).sum())} elements)"


def test_dwconv_layernorm_run_form_batched_and_ragged_equal_the_separate_kernels():
    """Round 6: at C = 128 / 256 / 512 / 1024 fo1_dwconv3x3_ln_bf16 walks runs of 8 pixels with the 3 x 3 window in registers.  Same bits as the
    separate kernels for stacked images (runs never cross a row or an image), for widths that are not a multiple of the run, and for the
    ragged entry (one workgroup column per image)."""
    from vlm_fo1_amd import lib as _L, ops
    with _L.use_ab():
        _L.load().fo1_dwconv_ln_set_form(2)         # the run form whatever the size (the product rule keeps small maps on the per-pixel form)
        try:
            _run_form_cases(ops)
        finally:
            _L.load().fo1_dwconv_ln_set_form(1)
    # product rule, a map large enough to take the run form by itself
    B, H, W, C = 2, 160, 120, 256
    x = torch.randn(B * H * W, C).to(BF).cuda()
    w9, b = (torch.randn(9, C) * 0.2).to(BF).cuda(), (torch.randn(C) * 0.1).to(BF).cuda()
    lw, lb = (1 + 0.1 * torch.randn(C)).to(BF).cuda(), (0.1 * torch.randn(C)).to(BF).cuda()
    y_ref = ops.dwconv3x3_res(x, w9, b, H, W, batch=B)
    y, h = ops.dwconv3x3_res_ln(x, w9, b, H, W, lw, lb, 1e-5, batch=B)
    assert torch.equal(y, y_ref) and torch.equal(h, ops.layernorm(y_ref, lw, lb, 1e-5))


def _run_form_cases(ops):
    torch.manual_seed(32)
    for (B, H, W, C) in [(3, 5, 19, 128), (2, 9, 8, 256), (3, 6, 33, 512), (2, 11, 7, 1024), (4, 1, 1, 1024), (1, 2, 64, 256)]:
        x = torch.randn(B * H * W, C).to(BF).cuda()
        w9 = (torch.randn(9, C) * 0.2).to(BF).cuda()
        b = (torch.randn(C) * 0.1).to(BF).cuda()
        lw = (1 + 0.1 * torch.randn(C)).to(BF).cuda()
        lb = (0.1 * torch.randn(C)).to(BF).cuda()
        y_ref = ops.dwconv3x3_res(x, w9, b, H, W, batch=B)
        h_ref = ops.layernorm(y_ref, lw, lb, 1e-5)
        y, h = ops.dwconv3x3_res_ln(x, w9, b, H, W, lw, lb, 1e-5, batch=B)
        assert torch.equal(y, y_ref) and torch.equal(h, h_ref), f"batch {B} x {H}x{W}x{C}"
    # ragged pack: three images of different sizes, per-image results
    C = 512
    sizes = [(7, 13), (3, 40), (10, 9)]
    xs = [torch.randn(h_ * w_, C).to(BF).cuda() for h_, w_ in sizes]
    w9 = (torch.randn(9, C) * 0.2).to(BF).cuda()
    b = (torch.randn(C) * 0.1).to(BF).cuda()
    lw = (1 + 0.1 * torch.randn(C)).to(BF).cuda()
    lb = (0.1 * torch.randn(C)).to(BF).cuda()
    rows, r0 = [], 0
    for h_, w_ in sizes:
        rows.append((r0, h_, w_, r0, h_, w_))
        r0 += h_ * w_
    sg = ops.ImgSegs(rows, "cuda", max(h_ * w_ for h_, w_ in sizes), r0, max(h_ * w_ for h_, w_ in sizes), r0)
    y, h = ops.dwconv3x3_res_ln_var(torch.cat(xs, 0), w9, b, sg, lw, lb, 1e-5)
    r = 0
    for (h_, w_), xi in zip(sizes, xs):
        y1, h1 = ops.dwconv3x3_res_ln(xi, w9, b, h_, w_, lw, lb, 1e-5)
        n = h_ * w_
        assert torch.equal(y[r:r + n], y1) and torch.equal(h[r:r + n], h1), f"ragged image {h_}x{w_}"
        r += n


def test_batched_spatial_ops_equal_per_image_calls():
    """ABI 2: `batch` same-size images stacked along the rows.  Every spatial kernel must give, for image b, exactly what the
    one-image call gives (neighbourhoods / windows / channel-attention statistics never cross an image boundary)."""
    from vlm_fo1_amd import ops
    torch.manual_seed(3)
    B, H, W, C = 3, 17, 23, 64
    x = torch.randn(B * H * W, C).to(BF).cuda()
    w9, b9 = (torch.randn(9, C) * 0.2).to(BF).cuda(), (torch.randn(C) * 0.1).to(BF).cuda()
    lw, lb = (1 + 0.1 * torch.randn(C)).to(BF).cuda(), (0.1 * torch.randn(C)).to(BF).cuda()
    per = lambda t, n: [t[i * n:(i + 1) * n].contiguous() for i in range(B)]
    yb, hb = ops.dwconv3x3_res_ln(x, w9, b9, H, W, lw, lb, 1e-5, batch=B)
    assert torch.equal(ops.dwconv3x3_res(x, w9, b9, H, W, batch=B), yb)
    for i, xi in enumerate(per(x, H * W)):
        y1, h1 = ops.dwconv3x3_res_ln(xi, w9, b9, H, W, lw, lb, 1e-5)
        assert torch.equal(y1, yb[i * H * W:(i + 1) * H * W]) and torch.equal(h1, hb[i * H * W:(i + 1) * H * W]), f"dwconv_ln image {i}"
    for (k, s, p) in ((3, 2, 1), (3, 1, 1), (7, 4, 3)):
        colb, Ho, Wo = ops.im2col(x, H, W, k, k, s, p, batch=B)
        for i, xi in enumerate(per(x, H * W)):
            c1, _, _ = ops.im2col(xi, H, W, k, k, s, p)
            assert torch.equal(c1, colb[i * Ho * Wo:(i + 1) * Ho * Wo]), f"im2col k{k} image {i}"
    ws = 12
    xw = ops.window_partition(x, H, W, ws, batch=B)
    nrow = xw.shape[0] // B
    yw = torch.randn_like(xw)
    back = ops.window_reverse_add(yw, x, H, W, ws, batch=B)
    for i, xi in enumerate(per(x, H * W)):
        assert torch.equal(ops.window_partition(xi, H, W, ws), xw[i * nrow:(i + 1) * nrow]), f"window_partition image {i}"
        assert torch.equal(ops.window_reverse_add(yw[i * nrow:(i + 1) * nrow].contiguous(), xi, H, W, ws), back[i * H * W:(i + 1) * H * W])
    qkv = torch.randn(B * H * W, 3 * C).to(BF).cuda()
    ab = ops.channel_attention(qkv, C, batch=B)
    for i, qi in enumerate(per(qkv, H * W)):
        assert torch.equal(ops.channel_attention(qi, C), ab[i * H * W:(i + 1) * H * W]), f"channel attention image {i}"
    Co = 16
    src = torch.randn(B * H * W, 4 * Co).to(BF).cuda()
    ps = ops.pixel_shuffle2(src, H, W, Co, batch=B)
    mp = ops.maxpool2(x, H, W, batch=B)
    for i in range(B):
        assert torch.equal(ops.pixel_shuffle2(src[i * H * W:(i + 1) * H * W].contiguous(), H, W, Co), ps[i * 4 * H * W:(i + 1) * 4 * H * W])
        n2 = (H // 2) * (W // 2)
        assert torch.equal(ops.maxpool2(x[i * H * W:(i + 1) * H * W].contiguous(), H, W), mp[i * n2:(i + 1) * n2])
    img = torch.randn(B, 3, H, W).cuda()
    hb8 = ops.nchw_to_hwc8(img)
    for i in range(B):
        assert torch.equal(ops.nchw_to_hwc8(img[i].contiguous()), hb8[i * H * W:(i + 1) * H * W])
