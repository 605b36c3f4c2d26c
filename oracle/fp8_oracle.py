"""CPU restatement of the fp8 (OCP e4m3fn) linear used for BASELINE configs[4] — TEST INFRASTRUCTURE ONLY (imported by tests/ only).

The reference has no fp8 path (it loads and runs bf16 everywhere, vlm_fo1/model/builder.py:40-46), so there is nothing in
/root/reference to pin this against: PARITY UNPINNED in the sense of the task's oracle rule.  What it is anchored on instead:
  * the e4m3fn conversion is PyTorch's own (`Tensor.to(torch.float8_e4m3fn)`: round to nearest even, subnormals, no infinities),
    checked against a hand-derived known-answer table in tests/test_oracle_fp8.py (OCP 8-bit floating point spec, e4m3: bias 7,
    max 448 = 0x7E, smallest subnormal 2^-9 = 0x01);
  * the linear is the plain definition: C = (dequant(Aq) dequant(Wq)^T) with per-row / per-output-channel scales, fp32 (here
    fp64) accumulation, followed by the rounding points of the bf16 GEMM epilogue (modeling_qwen2_5_vl.py nn.Linear call sites:
    bias -> bf16 -> act -> bf16 -> + residual -> bf16)."""
import numpy as np
import torch

E4M3_MAX = 448.0


def quantize_rows_e4m3(x: torch.Tensor):
    """x: [M, K] (any float dtype holding bf16-representable values).  Returns (q uint8 [M, K], scales fp32 [M]) — the same
    arithmetic as fo1_quantize_rows_e4m3: scale = absmax / 448 in fp32 (1 for a zero row), q = e4m3(clamp(x / scale))."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / torch.tensor(E4M3_MAX, dtype=torch.float32), torch.ones_like(amax))
    y = (xf / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8)
    return q, scale


def dequant(q: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float()


def _rb(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def gemm_fp8(aq, sa, wq, sw, bias=None, residual=None, act=0):
    """fp32 result (bf16-valued) of fo1_gemm_fp8.  act: 0 none, 1 GELU(erf), 2 SiLU, 3 interleaved SwiGLU (16-row groups)."""
    acc = (dequant(aq).double() @ dequant(wq).double().T).float() * (sa.float()[:, None] * sw.float()[None, :])
    if act == 3:
        M, N = acc.shape
        if bias is not None:
            acc = acc + bias.float()[None, :]
        g = acc.view(M, N // 32, 2, 16)
        gate, up = _rb(g[:, :, 0, :]), _rb(g[:, :, 1, :])
        return _rb(_rb(torch.nn.functional.silu(gate)) * up).reshape(M, N // 2)
    v = acc if bias is None else acc + bias.float()[None, :]
    v = _rb(v)
    if act == 1:
        v = _rb(torch.nn.functional.gelu(v))
    elif act == 2:
        v = _rb(torch.nn.functional.silu(v))
    if residual is not None:
        v = _rb(v + residual.float())
    return v
