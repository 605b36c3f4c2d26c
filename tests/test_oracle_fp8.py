"""fp8 oracle (oracle/fp8_oracle.py) against hand-derived e4m3fn known answers (OCP FP8: 1-4-3, bias 7, max 448, no inf) — CPU."""
import torch

from oracle import fp8_oracle as F


def test_e4m3_known_answers():
    # value -> byte.  exponent field e, mantissa m: (1 + m/8) * 2^(e-7); subnormals m/8 * 2^-6
    table = {0.0: 0x00, 1.0: 0x38, -1.0: 0xB8, 1.125: 0x39, 448.0: 0x7E, -448.0: 0xFE, 2.0 ** -9: 0x01, 2.0 ** -6: 0x08,
             0.875 * 2.0 ** -6: 0x07, 240.0: 0x77, 1.0625: 0x38,            # halfway 1.0 / 1.125 -> even mantissa (1.0)
             1.1875: 0x3A,                                                   # halfway 1.125 / 1.25 -> even (1.25)
             2.0 ** -10: 0x00,                                               # halfway 0 / 2^-9 -> even (0)
             3.0 * 2.0 ** -10: 0x02}                                         # halfway 2^-9 / 2^-8 -> even (2^-8)
    x = torch.tensor(list(table.keys()), dtype=torch.float32)
    got = x.to(torch.float8_e4m3fn).view(torch.uint8).tolist()
    assert got == list(table.values())


def test_quantize_rows_scales_and_bytes():
    x = torch.tensor([[0.0] * 8, [448.0, -224.0, 1.0, 0.5, 0.0, 0.0, 0.0, 0.0], [3.0, -1.5, 0.75, 0.0, 0.0, 0.0, 0.0, 6.0]])
    q, s = F.quantize_rows_e4m3(x)
    assert torch.equal(s, torch.tensor([1.0, 1.0, 6.0], dtype=torch.float32) / torch.tensor([1.0, 1.0, 448.0], dtype=torch.float32))
    assert q[0].tolist() == [0] * 8
    assert q[1, :4].tolist() == [0x7E, 0xF6, 0x38, 0x30]
    assert q[2, 7].item() == 0x7E and q[2, 0].item() == 0x76 and q[2, 1].item() == 0xEE     # 224 = 1.75 * 2^7, -112 = -1.75 * 2^6
    back = F.dequant(q) * s[:, None]
    assert torch.allclose(back, x, rtol=2 ** -4, atol=0)


def test_gemm_definition_small():
    torch.manual_seed(0)
    a = torch.randn(5, 128).bfloat16().float()
    w = torch.randn(8, 128).bfloat16().float()
    aq, sa = F.quantize_rows_e4m3(a)
    wq, sw = F.quantize_rows_e4m3(w)
    got = F.gemm_fp8(aq, sa, wq, sw)
    want = ((F.dequant(aq) * sa[:, None]) @ (F.dequant(wq) * sw[:, None]).T).bfloat16().float()
    assert torch.allclose(got, want, rtol=2 ** -7, atol=1e-3)
    exact = a @ w.T
    assert torch.nn.functional.cosine_similarity(got.flatten(), exact.flatten(), dim=0) > 0.998
