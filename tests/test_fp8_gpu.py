"""fp8 (e4m3) linear on the GPU against oracle/fp8_oracle.py: the quantiser bit for bit, the GEMM at fp32-accumulation tolerance
(BASELINE configs[4]; the reference has no fp8 path — see the oracle's header)."""
import pytest
import torch

from oracle import fp8_oracle as F

pytestmark = pytest.mark.gpu


def _ops():
    from vlm_fo1_amd import ops
    return ops


def _bf(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16()


def test_quantize_rows_bit_exact():
    ops = _ops()
    x = _bf(300, 1280, seed=1, scale=3.0)
    x[7] = 0                                   # all-zero row -> scale 1, bytes 0
    x[11, 5] = 3.0e4                           # one outlier: the rest of the row lands in the subnormals
    x[12] = x[12] * 1e-3
    q, s = ops.quantize_rows_fp8(x.cuda())
    q_ref, s_ref = F.quantize_rows_e4m3(x.float())
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)


def test_quantize_rows_long_row():
    """K > 12288: past the chunks a thread keeps in registers (the tail is re-read)."""
    ops = _ops()
    x = _bf(40, 13312, seed=12, scale=2.0)
    q, s = ops.quantize_rows_fp8(x.cuda())
    q_ref, s_ref = F.quantize_rows_e4m3(x.float())
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu(), q_ref)


def test_quantize_rows_strided_view():
    ops = _ops()
    x = _bf(64, 512, seed=2)
    xv = x.cuda()[:, 128:384]                  # ld 512, K 256
    q, s = ops.quantize_rows_fp8(xv)
    q_ref, s_ref = F.quantize_rows_e4m3(x[:, 128:384].float())
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu(), q_ref)


def _check(got, want, rare=2e-3, mag=None):
    """mag: magnitude the roundings happen at (|value before the residual| + |residual|: a cancelling residual leaves the ulp of the
    large intermediate in a small result)."""
    got, want = got.float().cpu(), want.float()
    mag = want.abs() if mag is None else mag.float()
    d = (got - want).abs()
    tol = mag * 2 ** -7 + 1e-3                 # 1 bf16 ulp (fp32 sum order moves a rounding now and then) + small abs
    bad = (d > tol).float().mean().item()
    assert bad <= rare, f"{bad:.2e} of the elements beyond 1 bf16 ulp (max diff {d.max().item():.3e})"
    assert (d <= mag * 2 ** -5 + 1e-2).all(), f"max diff {d.max().item():.3e}"


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 256), (1000, 1280, 1280), (517, 2564, 384), (2048, 1024, 3456)])
def test_gemm_fp8_identity_and_random(M, N, K):
    """A = selector rows (exactly representable, scale 1 after quantisation) against an asymmetric W: C must be W's dequantised rows —
    catches a row/column swap or a wrong k order; then random operands against the oracle."""
    ops = _ops()
    w = _bf(N, K, seed=3, scale=0.5)
    wq, sw = ops.quantize_rows_fp8(w.cuda())
    a = torch.zeros(M, K)
    cols = (torch.arange(M) * 7 + 3) % K
    a[torch.arange(M), cols] = 448.0           # absmax 448 -> scale 1, byte 0x7E
    aq, sa = ops.quantize_rows_fp8(a.bfloat16().cuda())
    got = ops.gemm_fp8(aq, sa, ops.Fp8Weight(wq, sw))
    want = (F.dequant(wq.cpu()) * sw.cpu()[:, None]).T[cols] * 448.0
    _check(got, want.bfloat16().float(), rare=0.0)
    a = _bf(M, K, seed=4)
    aq, sa = ops.quantize_rows_fp8(a.cuda())
    got = ops.gemm_fp8(aq, sa, ops.Fp8Weight(wq, sw))
    want = F.gemm_fp8(aq.cpu(), sa.cpu(), wq.cpu(), sw.cpu())
    _check(got, want)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_fp8_epilogues(act):
    ops = _ops()
    M, N, K = 700, 1536, 512
    a, w = _bf(M, K, seed=5), _bf(N, K, seed=6, scale=0.1)
    bias = _bf(N, seed=7)
    res = _bf(M, N, seed=8) if act != 3 else None
    aq, sa = ops.quantize_rows_fp8(a.cuda())
    wq, sw = ops.quantize_rows_fp8(w.cuda())
    got = ops.gemm_fp8(aq, sa, ops.Fp8Weight(wq, sw), bias.cuda(), res.cuda() if res is not None else None, act)
    want = F.gemm_fp8(aq.cpu(), sa.cpu(), wq.cpu(), sw.cpu(), bias, res, act)
    mag = None
    if res is not None:
        mag = F.gemm_fp8(aq.cpu(), sa.cpu(), wq.cpu(), sw.cpu(), bias, None, act).abs() + res.float().abs()
    _check(got, want, rare=1e-2 if act == 3 else 5e-3, mag=mag)


@pytest.mark.parametrize("M,D", [(300, 1280), (1000, 2048), (37, 4096)])
def test_rmsnorm_quant_equals_rmsnorm_then_quantize(M, D):
    ops = _ops()
    x = _bf(M, D, seed=20, scale=2.0).cuda()
    x[5] = 0
    w = (1.0 + 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(21))).bfloat16().cuda()
    q, s = ops.rmsnorm_quant_fp8(x, w, 1e-6)
    q2, s2 = ops.quantize_rows_fp8(ops.rmsnorm(x, w, 1e-6))
    assert torch.equal(s, s2) and torch.equal(q, q2)


def test_norm_linear_is_the_two_calls_in_bf16_and_the_fused_pair_in_fp8():
    ops = _ops()
    M, N, K = 1024, 2560, 2048
    x, w = _bf(M, K, seed=22).cuda(), _bf(N, K, seed=23, scale=0.05).cuda()
    nw, bias = _bf(K, seed=24).cuda(), _bf(N, seed=25).cuda()
    ref = ops.gemm(ops.rmsnorm(x, nw, 1e-6), w, bias)
    assert torch.equal(ops.norm_linear(x, nw, 1e-6, w, bias), ref)
    try:
        assert ops.register_fp8_weight(w)
        got = ops.norm_linear(x, nw, 1e-6, w, bias)
        two = ops.gemm(ops.rmsnorm(x, nw, 1e-6), w, bias)          # norm, stand-alone quantiser, fp8 product
    finally:
        ops.clear_fp8_weights()
    assert torch.equal(got, two)


def test_gemm_routes_registered_weights_and_tracks_bf16():
    ops = _ops()
    M, N, K = 2000, 2560, 2048
    a, w = _bf(M, K, seed=9).cuda(), _bf(N, K, seed=10, scale=0.05).cuda()
    ref = ops.gemm(a, w).float()
    try:
        assert ops.register_fp8_weight(w)
        got = ops.gemm(a, w).float()
        small = ops.gemm(a[:64], w).float()     # below FP8_MIN_ROWS: the bf16 kernel
    finally:
        ops.clear_fp8_weights()
    assert torch.equal(small, ref[:64])
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    rel = ((got - ref).norm() / ref.norm()).item()
    print(f"fp8 vs bf16 linear ({M}x{N}x{K}, gaussian operands): cos {cos:.5f}, rel {rel:.4f}")
    assert cos > 0.999 and rel < 0.04
