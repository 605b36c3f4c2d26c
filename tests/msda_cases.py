"""Seeded inputs of the multi-scale deformable attention cases (shared by tests/golden/make_msda_golden.py, which evaluates them
with the reference's own ms_deform_attn_core_pytorch, and by the oracle / GPU tests, which regenerate them bit for bit: the CPU
generator is deterministic for a given torch build, and both boxes run the same image)."""
import torch

CASES = {
    # UPN decoder geometry (reference configs/upn_large.py: 8 heads x 32 channels, 5 levels, 4 points; 900 queries there, 300 here),
    # locations drawn from [-0.15, 1.15] so that the zero-padding border rules are exercised
    "upn_decoder": dict(N=1, M=8, D=32, Lq=300, shapes=[(25, 34), (13, 17), (7, 9), (4, 5), (2, 3)], P=4, lo=-0.15, hi=1.15),
    # encoder geometry: one query per pyramid position
    "upn_encoder": dict(N=1, M=8, D=32, Lq=312, shapes=[(13, 17), (7, 9), (4, 5), (2, 3), (1, 2)], P=4, lo=0.0, hi=1.0),
    # D not a multiple of 4, 3 heads, 2 images, degenerate levels (1 row / 1 column)
    "ragged": dict(N=2, M=3, D=7, Lq=37, shapes=[(9, 5), (1, 6), (4, 1)], P=3, lo=-0.3, hi=1.3),
}
SEED = 20250926


def draw(name):
    """-> (value [N,S,M,D], shapes [(H,W)], level_start [L], loc [N,Lq,M,L,P,2], weight [N,Lq,M,L,P]) fp32, CPU."""
    c = CASES[name]
    gen = torch.Generator().manual_seed(SEED + sorted(CASES).index(name))
    shapes = c["shapes"]
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.rand(c["N"], S, c["M"], c["D"], generator=gen) * 2 - 1
    loc = torch.rand(c["N"], c["Lq"], c["M"], L, c["P"], 2, generator=gen) * (c["hi"] - c["lo"]) + c["lo"]
    w = torch.rand(c["N"], c["Lq"], c["M"], L, c["P"], generator=gen) + 1e-5
    w = w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    start = [0]
    for h, ww in shapes[:-1]:
        start.append(start[-1] + h * ww)
    return value, shapes, start, loc, w


def reference_test_inputs():
    """The inputs of the reference's own ops/test.py (lines 21-31, 34-59): N, M, D = 1, 2, 2; Lq, L, P = 2, 2, 2; shapes (6, 4), (3, 2);
    torch.manual_seed(3); the double check draws first, the float check second.  -> {tag: (value, shapes, level_start, loc, weight)}"""
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    state = torch.get_rng_state()
    torch.manual_seed(3)
    out = {}
    for tag, dt in (("test_py_f64", torch.float64), ("test_py_f32", torch.float32)):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        out[tag] = (value.to(dt), shapes, [0, 24], loc.to(dt), w.to(dt))
    torch.set_rng_state(state)
    return out
