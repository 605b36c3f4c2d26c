"""Thin torch-tensor front ends for the C-ABI ops in libfo1hip.so (include/fo1.h).

torch is only the allocator / stream owner here: every function validates layouts, allocates the
output with torch.empty and enqueues the HIP kernel(s) on the current stream.  No function has a
PyTorch fallback; a missing library or a CPU tensor raises."""
from __future__ import annotations

import math
import itertools
import os
import threading
from typing import Optional, Sequence

import ctypes

import torch

from . import lib as _L

ACT_NONE, ACT_GELU, ACT_SILU, ACT_SWIGLU16, ACT_RELU = 0, 1, 2, 3, 5


def _chk(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    if t.device.type != "cuda":
        raise _L.Fo1Error(f"{name}: expected a HIP device tensor, got {t.device}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _rows(t: torch.Tensor, name: str):
    """2-D view [M, D] with unit inner stride -> (ptr, ld, M, D)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D row-major tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), t.stride(0), t.shape[0], t.shape[1]


def _stream():
    return _L.current_stream_ptr()


# ---- caller-owned scratch (the library never allocates) ----------------------------------------------------------------
# One buffer per (kind, device, owner, size class).  `owner` is set by an engine for the duration of its device work
# (`workspace_scope`): replicas running on different streams — and the hipGraphs they captured, which hold raw pointers —
# must never share partial-sum scratch.  Outside any scope the key falls back to the current stream.  Buffers are never
# freed or resized in place: a captured graph may still reference them.
_ws_pool = {}
_ws_tls = threading.local()


class _CaptureLock:
    """hipGraph capture (and the eager warm-up pass before it) is exclusive; graph replays from other request threads are
    shared.  Concurrent stream captures from several host threads crash inside the runtime (segfault in capture_end on
    ROCm 7.2); replays only hold the lock for the host-side launch call, so the GPU work still overlaps."""

    def __init__(self):
        self._cond = threading.Condition()
        self._readers = 0
        self._writer = False

    class _Side:
        def __init__(self, enter, leave):
            self._enter, self._leave = enter, leave

        def __enter__(self):
            self._enter()

        def __exit__(self, *exc):
            self._leave()
            return False

    def _r_in(self):
        with self._cond:
            while self._writer:
                self._cond.wait()
            self._readers += 1

    def _r_out(self):
        with self._cond:
            self._readers -= 1
            if self._readers == 0:
                self._cond.notify_all()

    def _w_in(self):
        with self._cond:
            while self._writer or self._readers:
                self._cond.wait()
            self._writer = True

    def _w_out(self):
        with self._cond:
            self._writer = False
            self._cond.notify_all()

    def replay(self):
        return self._Side(self._r_in, self._r_out)

    def capture(self):
        return self._Side(self._w_in, self._w_out)


graph_lock = _CaptureLock()


class workspace_scope:
    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        self.prev = getattr(_ws_tls, "owner", None)
        _ws_tls.owner = self.owner
        return self

    def __exit__(self, *exc):
        _ws_tls.owner = self.prev
        return False


class keep_scope:
    """While active on this thread, host objects that own device index tables handed to launches (plans evicted from bounded caches:
    Conv3x3Plan) are appended to `keep` — the list a packed pass returns with its result and a captured hipGraph of that pass is stored
    with, so the tables outlive their cache entry for as long as a graph can replay launches that point at them (ADVICE r3's rule for
    the ragged tower plans, applied to the uniform paths' convolution plans)."""

    def __init__(self, keep: list):
        self.keep = keep

    def __enter__(self):
        self.prev = getattr(_ws_tls, "keep", None)
        _ws_tls.keep = self.keep
        return self

    def __exit__(self, *exc):
        _ws_tls.keep = self.prev
        return False


def keep_alive(obj) -> None:
    keep = getattr(_ws_tls, "keep", None)
    if keep is not None and not any(o is obj for o in keep):
        keep.append(obj)


def _workspace(kind: str, device, nbytes: int) -> torch.Tensor:
    owner = getattr(_ws_tls, "owner", None)
    who = ("owner", owner) if owner is not None else ("stream", torch.cuda.current_stream().cuda_stream)
    size = 1 << max(12, int(nbytes - 1).bit_length())      # power-of-two size classes
    key = (kind, device, who, size)
    ws = _ws_pool.get(key)
    if ws is None:
        ws = torch.zeros(size, dtype=torch.uint8, device=device)   # zeroed once: decode attention keeps arrival counters in its tail
        _ws_pool[key] = ws
    return ws


_zero_framed_kinds: dict = {}
_zero_framed_lock = threading.Lock()
ZERO_FRAMED_MAX = 48       # distinct persistent zero-framed buffers per process (each lives as long as its owner: a captured pass holds its raw pointer)


def zero_framed(key: tuple, rows: int, cols: int, device) -> torch.Tensor:
    """[rows, cols] bf16 scratch whose elements the caller NEVER writes are zero — the frame of a zero-padded convolution input, the padding
    columns of a V^T map.  `key` must determine which elements get written (a plan's serial number, a shape): the buffer is then zeroed once,
    at its first use, instead of by a fill launch per pass (8 fills = 0.25 ms of the 25-image pass).  One buffer per (key, engine replica /
    stream) from the scratch pool; past ZERO_FRAMED_MAX distinct keys (a server seeing ever new geometries) a fresh torch.zeros, as before."""
    if os.environ.get("FO1_ZERO_FRAMED", "1") != "0":
        with _zero_framed_lock:
            kind = _zero_framed_kinds.get(key)
            if kind is None and len(_zero_framed_kinds) < ZERO_FRAMED_MAX:
                kind = _zero_framed_kinds[key] = f"zero_framed_{len(_zero_framed_kinds)}"
        if kind is not None:
            nbytes = rows * cols * 2
            return _workspace(kind, device, nbytes)[:nbytes].view(torch.bfloat16).view(rows, cols)
    return torch.zeros(rows, cols, dtype=torch.bfloat16, device=device)


def release_workspaces(owner) -> int:
    """Drop every scratch buffer keyed by `owner` (an engine / decoder that is gone: its graphs, the only other holders of the raw
    pointers, went with it).  Returns the number of buffers released."""
    dead = [k for k in _ws_pool if k[2] == ("owner", owner)]
    for k in dead:
        del _ws_pool[k]
    return len(dead)


def new_owner(holder):
    """A scratch-pool owner token whose buffers are released when `holder` (the engine / decoder using it) is collected: repeated
    engine or replica creation must not leak device memory."""
    import weakref
    token = object()
    weakref.finalize(holder, release_workspaces, token)
    return token


def _gemm_workspace(device) -> torch.Tensor:
    """fp32 scratch for split-K partials, 64 MiB."""
    return _workspace("gemm", device, 64 * 1024 * 1024)


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, out: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T)  (nn.Linear semantics, fo1_gemm_bf16)."""
    _chk(a, "a"); _chk(w, "w")
    pa, lda, M, K = _rows(a, "a")
    pw, ldw, N, K2 = _rows(w, "w")
    if K != K2:
        raise ValueError(f"gemm: K mismatch {K} vs {K2}")
    n_out = N // 2 if act == ACT_SWIGLU16 else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    po, ldc, Mo, No = _rows(out, "out")
    assert (Mo, No) == (M, n_out)
    pr, ldr = (None, 0)
    if residual is not None:
        _chk(residual, "residual")
        pr, ldr, Mr, Nr = _rows(residual, "residual")
        assert (Mr, Nr) == (M, N)
    if bias is not None:
        _chk(bias, "bias")
        assert bias.numel() == N and bias.is_contiguous()
    if _fp8_weights and not out_f32 and M >= FP8_MIN_ROWS:
        fw = _fp8_weights.get((pw, N, K))
        if fw is not None:
            aq, sa = quantize_rows_fp8(a)
            return gemm_fp8(aq, sa, fw, bias, residual, act, out)
    ws = _gemm_workspace(a.device)
    rc = _L.load().fo1_gemm_bf16_ws(pa, lda, pw, ldw, bias.data_ptr() if bias is not None else None, pr, ldr, po, ldc,
                                    M, N, K, act, 1 if out_f32 else 0, ws.data_ptr(), ws.numel(), _stream())
    _L.check(rc, "fo1_gemm_bf16_ws")
    return out


# ---- fp8 linear (BASELINE configs[4]) ---------------------------------------------------------------------------------------
class Fp8Weight:
    """Per-output-channel e4m3 copy of an nn.Linear weight: q uint8 [N, K], scale fp32 [N].  `src` keeps the bf16 tensor it was made
    from alive: the routing table is keyed by that tensor's address, which must not be handed to another allocation meanwhile."""
    __slots__ = ("q", "scale", "src")

    def __init__(self, q: torch.Tensor, scale: torch.Tensor, src: Optional[torch.Tensor] = None):
        self.q, self.scale, self.src = q, scale, src


_fp8_weights = {}            # (data_ptr, N, K) of a registered bf16 weight -> Fp8Weight
FP8_MIN_ROWS = 512           # below this the 256 x 256 fp8 tile is mostly padding: the bf16 kernels run


def quantize_rows_fp8(x: torch.Tensor, q: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None):
    """(q uint8 [M, K], scales fp32 [M]) = row-wise e4m3 quantisation of a bf16 matrix (fo1_quantize_rows_e4m3)."""
    _chk(x, "x")
    px, ldx, M, K = _rows(x, "x")
    if q is None:
        q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    if scales is None:
        scales = torch.empty(M, dtype=torch.float32, device=x.device)
    _L.check(_L.load().fo1_quantize_rows_e4m3(px, ldx, M, K, q.data_ptr(), q.stride(0), scales.data_ptr(), _stream()), "fo1_quantize_rows_e4m3")
    return q, scales


def gemm_fp8(aq: torch.Tensor, sa: torch.Tensor, w: Fp8Weight, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             act: int = ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = epilogue((aq @ w.q^T) * sa[:, None] * w.scale[None, :])  (fo1_gemm_fp8)."""
    M, K = aq.shape
    N = w.q.shape[0]
    if aq.dtype != torch.uint8 or w.q.dtype != torch.uint8 or w.q.shape[1] != K or not aq.is_cuda:
        raise ValueError("gemm_fp8: uint8 device operands with equal K")
    n_out = N // 2 if act == ACT_SWIGLU16 else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=aq.device)
    po, ldc, Mo, No = _rows(out, "out")
    assert (Mo, No) == (M, n_out)
    pr, ldr = (None, 0)
    if residual is not None:
        _chk(residual, "residual")
        pr, ldr, Mr, Nr = _rows(residual, "residual")
        assert (Mr, Nr) == (M, N)
    if bias is not None:
        _chk(bias, "bias")
        assert bias.numel() == N and bias.is_contiguous()
    rc = _L.load().fo1_gemm_fp8(aq.data_ptr(), aq.stride(0), sa.data_ptr(), w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr(),
                                bias.data_ptr() if bias is not None else None, pr, ldr, po, ldc, M, N, K, act, _stream())
    _L.check(rc, "fo1_gemm_fp8")
    return out


def rmsnorm_quant_fp8(x: torch.Tensor, weight: torch.Tensor, eps: float):
    """(q uint8 [M, D], scales fp32 [M]) = quantize_rows_fp8(rmsnorm(x, weight, eps)) in one launch (fo1_rmsnorm_quant_e4m3)."""
    _chk(x, "x"); _chk(weight, "weight")
    px, ldx, M, D = _rows(x, "x")
    q = torch.empty(M, D, dtype=torch.uint8, device=x.device)
    scales = torch.empty(M, dtype=torch.float32, device=x.device)
    _L.check(_L.load().fo1_rmsnorm_quant_e4m3(px, ldx, weight.data_ptr(), M, D, float(eps), q.data_ptr(), q.stride(0), scales.data_ptr(), _stream()),
             "fo1_rmsnorm_quant_e4m3")
    return q, scales


def norm_linear(x: torch.Tensor, norm_w: torch.Tensor, eps: float, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                act: int = ACT_NONE) -> torch.Tensor:
    """gemm(rmsnorm(x, norm_w, eps), w, bias, act=act): the RMSNorm -> nn.Linear pair of every ViT block / LLM layer
    (modeling_qwen2_5_vl.py:317-331, 1066-1090).  bf16: exactly those two calls.  When `w` is registered for fp8 and the product is
    large enough, the norm emits the e4m3 row + scale directly (no bf16 row, no separate quantiser launch)."""
    if _fp8_weights and x.shape[0] >= FP8_MIN_ROWS:
        pw, ldw, N, K = _rows(w, "w")
        fw = _fp8_weights.get((pw, N, K))
        if fw is not None:
            q, s = rmsnorm_quant_fp8(x, norm_w, eps)
            return gemm_fp8(q, s, fw, bias, None, act)
    return gemm(rmsnorm(x, norm_w, eps), w, bias, act=act)


def fp8_routed(w: torch.Tensor, rows: int) -> bool:
    """True when norm_linear / gemm would send a product of `rows` rows with this weight through the fp8 kernel."""
    if not _fp8_weights or rows < FP8_MIN_ROWS:
        return False
    pw, ldw, N, K = _rows(w, "w")
    return (pw, N, K) in _fp8_weights


def register_fp8_weight(w: torch.Tensor):
    """Quantise a bf16 [N, K] weight and let gemm() route large-M products with it through the fp8 kernel.  The bf16 tensor stays
    (decode and small-M products keep using it).  Returns the routing key (truthy), or None when the shape does not qualify
    (K % 128, N % 4)."""
    _chk(w, "w")
    pw, ldw, N, K = _rows(w, "w")
    if K % 128 or N % 4:
        return None
    q, s = quantize_rows_fp8(w)
    _fp8_weights[(pw, N, K)] = Fp8Weight(q, s, w)
    return (pw, N, K)


def clear_fp8_weights(keys=None) -> None:
    """Drop the given routing keys (an engine's own registrations), or every registration."""
    if keys is None:
        _fp8_weights.clear()
    else:
        for k in keys:
            _fp8_weights.pop(k, None)


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x"); _chk(weight, "weight")
    px, ldx, M, D = _rows(x, "x")
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    po, ldy, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_rmsnorm_bf16(px, ldx, weight.data_ptr(), po, ldy, M, D, float(eps), _stream()), "fo1_rmsnorm_bf16")
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x"); _chk(weight, "weight"); _chk(bias, "bias")
    px, ldx, M, D = _rows(x, "x")
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    po, ldy, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_layernorm_bf16(px, ldx, weight.data_ptr(), bias.data_ptr(), po, ldy, M, D, float(eps), _stream()),
             "fo1_layernorm_bf16")
    return out


def layernorm_rows(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out: torch.Tensor, out_rows: torch.Tensor) -> torch.Tensor:
    """LayerNorm whose output row m goes to row out_rows[m] (int32) of `out` — the padded map of an implicit-GEMM convolution
    (fo1_layernorm_rows_bf16; same arithmetic as layernorm)."""
    _chk(x, "x"); _chk(weight, "weight"); _chk(bias, "bias"); _chk(out, "out")
    px, ldx, M, D = _rows(x, "x")
    po, ldy, _, _ = _rows(out, "out")
    assert out_rows.dtype == torch.int32 and out_rows.is_contiguous() and out_rows.numel() == M and out_rows.device == x.device
    _L.check(_L.load().fo1_layernorm_rows_bf16(px, ldx, weight.data_ptr(), bias.data_ptr(), po, ldy, out_rows.data_ptr(), M, D, float(eps), _stream()),
             "fo1_layernorm_rows_bf16")
    return out


_conv_plan_serial = itertools.count()


class Conv3x3Plan:
    """Index tables of a 3x3 / pad 1 convolution run as an implicit GEMM (fo1_conv3x3_gemm_bf16) over images packed row-wise: the zero-padded
    layout (one pixel per side, common row pitch Wp = widest image + 2), `rowmap` int32 [sum H W] = padded row of every input pixel (where
    layernorm_rows writes it), `a_rows` uint32 [sum Ho Wo] = byte offset of every output pixel's top-left tap.  Host-built once per
    (sizes, stride, channels) and cached (index bookkeeping belongs on the host)."""

    def __init__(self, sizes, stride: int, cin: int, device):
        import numpy as np
        Wp = max(w for _, w in sizes) + 2
        rowmap, a_rows, out_hw = [], [], []
        pb = 0
        for (H, W) in sizes:
            Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            rowmap.append((pb + (yy + 1) * Wp + (xx + 1)).reshape(-1))
            oy, ox = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
            a_rows.append(((pb + oy * stride * Wp + ox * stride).astype(np.int64) * (cin * 2)).reshape(-1))
            out_hw.append((Ho, Wo))
            pb += (H + 2) * Wp
        ar = np.concatenate(a_rows)
        assert pb * cin * 2 < 2 ** 32, "padded map exceeds the 32-bit byte offsets of the implicit-GEMM convolution"
        self.Wp, self.cin, self.pad_rows, self.out_hw = Wp, cin, pb, out_hw
        self.serial = next(_conv_plan_serial)      # (identifies the padded map's written rows: ops.zero_framed)
        self.uniform = len(set(tuple(t) for t in sizes)) == 1
        self.M_in, self.M_out = sum(h * w for h, w in sizes), int(ar.shape[0])
        self.rowmap = torch.from_numpy(np.concatenate(rowmap).astype(np.int32)).to(device)
        self.a_rows = torch.from_numpy(ar.astype(np.uint32).view(np.int32)).to(device)      # (uint32 bit patterns in an int32 tensor)


def conv3x3_padded(pl: "Conv3x3Plan", device) -> torch.Tensor:
    """The zero-framed map layernorm_rows writes for plan `pl`: persistent for same-size batches (the frame stays zero from pass to pass, the
    interior is rewritten by every pass), a fresh zero fill for ragged packs (their geometry rarely repeats)."""
    if pl.uniform:
        return zero_framed(("conv3x3", pl.serial), pl.pad_rows, pl.cin, device)
    return torch.zeros(pl.pad_rows, pl.cin, dtype=torch.bfloat16, device=device)


_conv_plans: dict = {}
_conv_plans_lock = threading.Lock()      # the cache is shared by every engine replica / worker thread: ragged packs add ~7 plans per pass


def conv3x3_plan(sizes, stride: int, cin: int, device) -> Conv3x3Plan:
    key = (tuple((int(h), int(w)) for h, w in sizes), int(stride), int(cin), str(device))
    with _conv_plans_lock:
        pl = _conv_plans.get(key)
        if pl is None:
            while len(_conv_plans) >= 64:
                _conv_plans.pop(next(iter(_conv_plans)), None)
            pl = _conv_plans[key] = Conv3x3Plan(key[0], stride, cin, device)
    keep_alive(pl)       # a captured pass holds rowmap / a_rows by raw pointer: it keeps the plan past this cache's eviction
    return pl


def conv3x3_implicit_ok(M_out: int, cout: int, cin: int, k: int, pad: int) -> bool:
    """The implicit-GEMM form (no im2col matrix) applies to 3x3 / pad 1 convolutions over >= 64 power-of-two channels whose GEMM runs on the
    256 x 256 kernel anyway (then both forms give the same bits).  FO1_CONV_IMPLICIT=0 turns it off (A/B)."""
    return (os.environ.get("FO1_CONV_IMPLICIT", "1") != "0" and k == 3 and pad == 1 and cin >= 64 and (cin & (cin - 1)) == 0 and cout % 8 == 0
            and bool(_L.load().fo1_gemm_takes_big_tile(int(M_out), int(cout), 9 * int(cin))))


def conv3x3_implicit_for(sizes, stride: int, cout: int, cin: int, k: int, pad: int) -> bool:
    """conv3x3_implicit_ok for images of `sizes` [(H, W)] packed row-wise (uniform or ragged), including the 32-bit byte offsets of the padded
    map (Conv3x3Plan): False sends the caller to im2col + GEMM, the same bits."""
    if k != 3 or pad != 1:
        return False
    Wp = max(int(w) for _, w in sizes) + 2
    if sum(int(h) + 2 for h, _ in sizes) * Wp * int(cin) * 2 >= 2 ** 32:
        return False
    M_out = sum(((int(h) + 2 - 3) // stride + 1) * ((int(w) + 2 - 3) // stride + 1) for h, w in sizes)
    return conv3x3_implicit_ok(M_out, cout, cin, k, pad)


def conv3x3_gemm(xpad: torch.Tensor, plan: Conv3x3Plan, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """3x3 convolution of the zero-padded map `xpad` [plan.pad_rows, Cin] (layernorm_rows wrote it) as an implicit GEMM: -> [plan.M_out, Cout]."""
    _chk(xpad, "xpad"); _chk(w, "w")
    assert xpad.is_contiguous() and xpad.shape == (plan.pad_rows, plan.cin)
    pw, ldw, N, K = _rows(w, "w")
    assert K == 9 * plan.cin
    out = torch.empty(plan.M_out, N, dtype=torch.bfloat16, device=xpad.device)
    rc = _L.load().fo1_conv3x3_gemm_bf16(xpad.data_ptr(), plan.a_rows.data_ptr(), plan.Wp, plan.cin, pw, ldw, bias.data_ptr() if bias is not None else None,
                                         out.data_ptr(), N, plan.M_out, N, int(act), _stream())
    _L.check(rc, "fo1_conv3x3_gemm_bf16")
    return out


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[F, ...] gate and up -> [2F, ...] rows interleaved in 16-row groups [gate 16 | up 16 | ...], the weight/bias
    layout of the fused SwiGLU GEMM epilogue (act = ACT_SWIGLU16).  F must be a multiple of 16."""
    F = gate.shape[0]
    assert up.shape == gate.shape and F % 16 == 0
    g = gate.reshape(F // 16, 16, *gate.shape[1:])
    u = up.reshape(F // 16, 16, *up.shape[1:])
    return torch.stack([g, u], dim=1).reshape(2 * F, *gate.shape[1:]).contiguous()


def swiglu(gate_up: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(gate_up, "gate_up")
    p, ld, M, F2 = _rows(gate_up, "gate_up")
    F = F2 // 2
    if out is None:
        out = torch.empty(M, F, dtype=torch.bfloat16, device=gate_up.device)
    po, ldo, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_swiglu_bf16(p, ld, po, ldo, M, F, _stream()), "fo1_swiglu_bf16")
    return out


def bias_act(x: torch.Tensor, bias: Optional[torch.Tensor], act: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x")
    p, ld, M, D = _rows(x, "x")
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    po, ldo, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_bias_act_bf16(p, ld, bias.data_ptr() if bias is not None else None, po, ldo, M, D, act, _stream()),
             "fo1_bias_act_bf16")
    return out


def argmax(row: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(row, "row")
    assert row.is_contiguous()
    if out is None:
        out = torch.empty(1, dtype=torch.int32, device=row.device)
    sc = _workspace("argmax", row.device, 4096)   # per-owner: concurrent requests must not share the two-stage partials
    _L.check(_L.load().fo1_argmax_bf16(row.data_ptr(), row.numel(), out.data_ptr(), sc.data_ptr(), _stream()), "fo1_argmax_bf16")
    return out


def attention_decode(q: torch.Tensor, kcache: torch.Tensor, vtcache: torch.Tensor, kv_len_dev: torch.Tensor, max_kv_len: int,
                     n_q_heads: int, n_kv_heads: int, head_dim: int, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One new token (q [1, n_q_heads*head_dim]) against the cache; kv_len_dev = device int32 scalar holding the
    number of keys to attend (cache position + 1)."""
    _chk(q, "q"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    assert kv_len_dev.dtype == torch.int32 and kcache.dim() == 3 and q.shape[0] == 1
    need = _L.load().fo1_attention_decode_workspace_bytes(max_kv_len, n_kv_heads, head_dim)
    ws = _workspace("attn_decode", q.device, need)
    if out is None:
        out = torch.empty(1, n_q_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    rc = _L.load().fo1_attention_decode_bf16(q.data_ptr(), kcache.data_ptr(), kcache.stride(1), kcache.stride(0), pv, ldv, out.data_ptr(),
                                             kv_len_dev.data_ptr(), max_kv_len, n_q_heads, n_kv_heads, head_dim, float(scale),
                                             ws.data_ptr(), ws.numel(), _stream())
    _L.check(rc, "fo1_attention_decode_bf16")
    return out


def rope_llm(qkv: torch.Tensor, n_heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor,
             kcache: Optional[torch.Tensor] = None, k_first_head: int = 0, pos0: int = 0, col0: int = 0,
             dyn_state: Optional[torch.Tensor] = None) -> None:
    """In place on heads [0,n_heads) (q heads then k heads) of qkv [L, ld]; cos/sin bf16 [L, head_dim] — or, with
    dyn_state (device int32: [cache position, table row]), full per-position tables indexed on the device."""
    _chk(qkv, "qkv"); _chk(cos, "cos"); _chk(sin, "sin")
    p, ld, L, _ = _rows(qkv, "qkv")
    assert cos.shape[1] == head_dim and cos.is_contiguous() and sin.is_contiguous()
    assert dyn_state is not None or cos.shape[0] == L
    kc_ptr, kc_stride = None, 0
    if kcache is not None:
        _chk(kcache, "kcache")
        assert kcache.dim() == 3 and kcache.shape[2] == head_dim and kcache.stride(2) == 1 and kcache.stride(1) == head_dim
        kc_ptr, kc_stride = kcache.data_ptr(), kcache.stride(0)
    _L.check(_L.load().fo1_rope_llm_bf16(p, ld, col0, n_heads, head_dim, cos.data_ptr(), sin.data_ptr(), L, kc_ptr,
                                         k_first_head, kc_stride, pos0, dyn_state.data_ptr() if dyn_state is not None else None,
                                         _stream()), "fo1_rope_llm_bf16")


def rope_vit(qkv: torch.Tensor, n_heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor) -> None:
    _chk(qkv, "qkv"); _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
    p, ld, S, _ = _rows(qkv, "qkv")
    assert cos.shape == (S, head_dim // 2) and cos.is_contiguous() and sin.is_contiguous()
    _L.check(_L.load().fo1_rope_vit_bf16(p, ld, n_heads, head_dim, cos.data_ptr(), sin.data_ptr(), S, _stream()),
             "fo1_rope_vit_bf16")


def qkv_post_llm(qkv: torch.Tensor, n_q: int, n_kv: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor,
                 kcache: torch.Tensor, vtcache: torch.Tensor, pos0: int = 0) -> None:
    """Prefill, one launch: mRoPE on the q/k heads of qkv [L, (n_q+2 n_kv) hd] in place, K heads appended to
    kcache [n_kv, max_seq, hd] at pos0, V heads copied transposed into vtcache [n_kv*hd, max_seq] at column pos0."""
    _chk(qkv, "qkv"); _chk(cos, "cos"); _chk(sin, "sin"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    p, ld, L, _ = _rows(qkv, "qkv")
    assert cos.shape == (L, head_dim) and cos.is_contiguous() and sin.is_contiguous()
    assert kcache.dim() == 3 and kcache.shape[2] == head_dim and kcache.stride(2) == 1 and kcache.stride(1) == head_dim
    pv, ldv, Cv, _ = _rows(vtcache, "vtcache")
    assert Cv == n_kv * head_dim
    _L.check(_L.load().fo1_qkv_post_llm_bf16(p, ld, n_q, n_kv, head_dim, cos.data_ptr(), sin.data_ptr(), L, kcache.data_ptr(),
                                             kcache.stride(0), pv, ldv, pos0, _stream()), "fo1_qkv_post_llm_bf16")


def qkv_post_vit(qkv: torch.Tensor, n_heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor, vt: torch.Tensor) -> None:
    """ViT block, one launch: 2-D RoPE on the q/k heads of qkv [S, 3 d] in place + V -> vt [d, >= S]."""
    _chk(qkv, "qkv"); _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32); _chk(vt, "vt")
    p, ld, S, _ = _rows(qkv, "qkv")
    assert cos.shape == (S, head_dim // 2) and cos.is_contiguous() and sin.is_contiguous()
    pv, ldv, Cv, _ = _rows(vt, "vt")
    assert Cv == n_heads * head_dim
    _L.check(_L.load().fo1_qkv_post_vit_bf16(p, ld, n_heads, head_dim, cos.data_ptr(), sin.data_ptr(), S, pv, ldv, _stream()),
             "fo1_qkv_post_vit_bf16")


def qkv_fused_enabled() -> bool:
    """FO1_QKV_FUSED=0 turns the fused q/k/v epilogue off (A/B; both forms give the same bits)."""
    return os.environ.get("FO1_QKV_FUSED", "1") != "0"


def qkv_fused_for(M: int, N: int, K: int) -> bool:
    """Take the fused q/k/v form (qkv_proj_rope: always the 256 x 256 GEMM kernel) for a product that gemm() itself would run on that kernel
    — and only there: a row's fp32 sum order is the kernel's, so the fused and the two-launch form then agree bit for bit (also under the
    test build's tile pins, which fo1_gemm_takes_big_tile honours)."""
    return qkv_fused_enabled() and bool(_L.load().fo1_gemm_takes_big_tile(int(M), int(N), int(K)))


def qkv_proj_rope(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: int, n_q: int, n_kv: int, cos: torch.Tensor, sin: torch.Tensor,
                  kcache: Optional[torch.Tensor], pos0: int, vt: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/k/v projection + bias + RoPE + K-cache append + V^T write in one launch (fo1_qkv_proj_rope_bf16: the 256 x 256 GEMM with the fused
    epilogue).  mode 0 (LLM): w [(n_q + 2 n_kv) * 128, K], cos / sin bf16 [M, 128], kcache [n_kv, max_seq, 128]; returns [M, N] whose first
    n_q * 128 columns hold the rotated q heads (the k / v columns are not written).  mode 1 (ViT): w head-major [n_q * 256, K] (head_major_qkv),
    cos / sin fp32 [M, 40]; returns [M, n_q * 256] = per head [q 80 | k 80 | - | -] rotated.  vt: the V^T destination [heads * head_dim, ld]."""
    _chk(x, "x"); _chk(w, "w"); _chk(vt, "vt")
    px, ldx, M, K = _rows(x, "x")
    pw, ldw, N, Kw = _rows(w, "w")
    assert K == Kw
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    po, ldo, _, _ = _rows(out, "out")
    pv, ldv, _, _ = _rows(vt, "vt")
    if mode == 0:
        _chk(cos, "cos"); _chk(sin, "sin"); _chk(kcache, "kcache")
        assert cos.shape == (M, 128) and cos.is_contiguous() and sin.shape == (M, 128) and sin.is_contiguous()
        assert kcache.dim() == 3 and kcache.shape[2] == 128 and kcache.stride(2) == 1 and kcache.stride(1) == 128
        pk, ks = kcache.data_ptr(), kcache.stride(0)
    else:
        _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
        assert cos.shape == (M, 40) and cos.is_contiguous() and sin.shape == (M, 40) and sin.is_contiguous()
        pk, ks = None, 0
    rc = _L.load().fo1_qkv_proj_rope_bf16(px, ldx, pw, ldw, bias.data_ptr() if bias is not None else None, po, ldo, M, N, K, int(mode), int(n_q), int(n_kv),
                                          cos.data_ptr(), sin.data_ptr(), pk, ks, int(pos0), pv, ldv, _stream())
    _L.check(rc, "fo1_qkv_proj_rope_bf16")
    return out


def head_major_qkv(w: torch.Tensor, n_heads: int, head_dim: int = 80, tile: int = 256) -> torch.Tensor:
    """[3 * n_heads * head_dim, ...] rows ordered [q heads | k heads | v heads] -> [n_heads * tile, ...] rows ordered per head
    [q | k | v | zero pad] (qkv_proj_rope mode 1: one 256-column output tile per head).  Works for the weight [3 d, K] and the bias [3 d]."""
    d = n_heads * head_dim
    assert w.shape[0] == 3 * d and 3 * head_dim <= tile
    parts = w.reshape(3, n_heads, head_dim, *w.shape[1:]).transpose(0, 1)                     # [head, 3, head_dim, ...]
    out = torch.zeros(n_heads, tile, *w.shape[1:], dtype=w.dtype, device=w.device)
    out[:, :3 * head_dim] = parts.reshape(n_heads, 3 * head_dim, *w.shape[1:])
    return out.reshape(n_heads * tile, *w.shape[1:]).contiguous()


def transpose_into(src: torch.Tensor, dst: torch.Tensor, col0: int = 0, dyn_col0: Optional[torch.Tensor] = None) -> None:
    """dst[c, col0 + m] = src[m, c];  src [M, C] (C % 64 == 0), dst [C, >= col0 + M]; col0 read from the device int
    `dyn_col0` when given."""
    _chk(src, "src"); _chk(dst, "dst")
    p, ld, M, C = _rows(src, "src")
    pd, ldd, Cd, _ = _rows(dst, "dst")
    assert Cd == C
    _L.check(_L.load().fo1_transpose_bf16(p, ld, pd, ldd, col0, dyn_col0.data_ptr() if dyn_col0 is not None else None, M, C,
                                          _stream()), "fo1_transpose_bf16")


def gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 0.0,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode-step projection: out[M<=4, N] = epilogue(rmsnorm?(x) @ w^T) (fo1_gemv_bf16)."""
    _chk(x, "x"); _chk(w, "w")
    px, ldx, M, K = _rows(x, "x")
    pw, ldw, N, K2 = _rows(w, "w")
    assert K == K2 and M <= 4
    n_out = N // 2 if act == ACT_SWIGLU16 else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device)
    po, ldc, _, _ = _rows(out, "out")
    pr, ldr = (None, 0)
    if residual is not None:
        pr, ldr, _, _ = _rows(residual, "residual")
    rc = _L.load().fo1_gemv_bf16(px, ldx, pw, ldw, bias.data_ptr() if bias is not None else None, pr, ldr, po, ldc, M, N, K, act,
                                 norm_weight.data_ptr() if norm_weight is not None else None, float(norm_eps), _stream())
    _L.check(rc, "fo1_gemv_bf16")
    return out


# ---- batched decode (decode.hip) ------------------------------------------------------------------------------------------
GB_PLAIN, GB_SWIGLU, GB_QKV = 0, 1, 2


def gemv_batch(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               mode: int = GB_PLAIN, norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 0.0,
               out: Optional[torch.Tensor] = None, qkv: Optional[dict] = None) -> torch.Tensor:
    """Decode-step projection for M <= 32 sequences (fo1_gemv_batch_bf16).  qkv (mode GB_QKV): dict(n_q, n_kv, cos, sin, state,
    kcache [n_kv, rows, 128], vtcache [n_kv*128, rows]) — `out` then receives only the rotated q rows [M, n_q*128]."""
    _chk(x, "x"); _chk(w, "w")
    px, ldx, M, K = _rows(x, "x")
    pw, ldw, N, K2 = _rows(w, "w")
    assert K == K2 and M <= 32
    n_out = N // 2 if mode == GB_SWIGLU else (qkv["n_q"] * 128 if mode == GB_QKV else N)
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device)
    po, ldc, _, _ = _rows(out, "out")
    pr, ldr = (None, 0)
    if residual is not None:
        pr, ldr, _, _ = _rows(residual, "residual")
    if mode == GB_QKV:
        kc, vt = qkv["kcache"], qkv["vtcache"]
        _chk(kc, "kcache"); _chk(vt, "vtcache")
        assert kc.dim() == 3 and kc.stride(2) == 1 and kc.stride(1) == 128 and qkv["state"].dtype == torch.int32
        pv, ldv, _, _ = _rows(vt, "vtcache")
        extra = (qkv["n_q"], qkv["n_kv"], qkv["cos"].data_ptr(), qkv["sin"].data_ptr(), qkv["state"].data_ptr(), kc.data_ptr(), kc.stride(0), pv, ldv)
    else:
        extra = (0, 0, None, None, None, None, 0, None, 0)
    rc = _L.load().fo1_gemv_batch_bf16(px, ldx, pw, ldw, bias.data_ptr() if bias is not None else None, pr, ldr, po, ldc, M, N, K, mode,
                                       norm_weight.data_ptr() if norm_weight is not None else None, float(norm_eps), *extra, _stream())
    _L.check(rc, "fo1_gemv_batch_bf16")
    return out


def attention_decode_batch(q: torch.Tensor, kcache: torch.Tensor, vtcache: torch.Tensor, state: torch.Tensor, max_kv_len: int,
                           n_q_heads: int, n_kv_heads: int, head_dim: int, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B, n_q_heads*head_dim] (one new token per sequence) against the slots described by state int32 [B, 8]."""
    _chk(q, "q"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    assert state.dtype == torch.int32 and state.is_contiguous() and kcache.dim() == 3
    pq, ldq, B, _ = _rows(q, "q")
    need = _L.load().fo1_attention_decode_batch_workspace_bytes(max_kv_len, n_kv_heads, head_dim, B)
    ws = _workspace("attn_decode", q.device, need)
    if out is None:
        out = torch.empty(B, n_q_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    po, ldo, _, _ = _rows(out, "out")
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    rc = _L.load().fo1_attention_decode_batch_bf16(pq, ldq, kcache.data_ptr(), kcache.stride(1), kcache.stride(0), pv, ldv, po, ldo,
                                                   state.data_ptr(), B, max_kv_len, n_q_heads, n_kv_heads, head_dim, float(scale),
                                                   ws.data_ptr(), ws.numel(), _stream())
    _L.check(rc, "fo1_attention_decode_batch_bf16")
    return out


def attention_decode_batch_partials(q: torch.Tensor, kcache: torch.Tensor, vtcache: torch.Tensor, state: torch.Tensor, max_kv_len: int,
                                    n_q_heads: int, n_kv_heads: int, head_dim: int, scale: float):
    """The split-KV half of attention_decode_batch alone -> (partials workspace, floats per sequence, keys per chunk) for gemv_attn_combine."""
    _chk(q, "q"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    assert state.dtype == torch.int32 and state.is_contiguous() and kcache.dim() == 3
    pq, ldq, B, _ = _rows(q, "q")
    need = _L.load().fo1_attention_decode_batch_workspace_bytes(max_kv_len, n_kv_heads, head_dim, B)
    ws = _workspace("attn_decode", q.device, need)
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    chunk, stride = ctypes.c_int(0), ctypes.c_longlong(0)
    rc = _L.load().fo1_attention_decode_batch_partials_bf16(pq, ldq, kcache.data_ptr(), kcache.stride(1), kcache.stride(0), pv, ldv, state.data_ptr(), B,
                                                            max_kv_len, n_q_heads, n_kv_heads, head_dim, float(scale), ws.data_ptr(), ws.numel(),
                                                            ctypes.byref(chunk), ctypes.byref(stride), _stream())
    _L.check(rc, "fo1_attention_decode_batch_partials_bf16")
    return ws, stride.value, chunk.value


def gemv_attn_combine(part: torch.Tensor, part_seq_stride: int, state: torch.Tensor, kv_chunk: int, n_q_heads: int, n_kv_heads: int, w: torch.Tensor,
                      residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """o-projection of a decode step at <= 2 sequences, the attention combine in its prologue (fo1_gemv_attn_combine_bf16)."""
    _chk(w, "w")
    pw, ldw, N, K = _rows(w, "w")
    M = state.shape[0]
    assert K == n_q_heads * 128 and M <= 2 and state.dtype == torch.int32 and state.is_contiguous()
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=w.device)
    po, ldc, _, _ = _rows(out, "out")
    pr, ldr = (None, 0)
    if residual is not None:
        pr, ldr, _, _ = _rows(residual, "residual")
    rc = _L.load().fo1_gemv_attn_combine_bf16(part.data_ptr(), int(part_seq_stride), state.data_ptr(), int(kv_chunk), n_q_heads, n_kv_heads, pw, ldw, pr, ldr,
                                              po, ldc, M, N, _stream())
    _L.check(rc, "fo1_gemv_attn_combine_bf16")
    return out


# ---- decode pool: 64 / 128 sequence slots per weight stream (llm.DecodePool) ---------------------------------------------------------
def pool_qkv_post(qkv: torch.Tensor, n_q: int, n_kv: int, head_dim: int, cos_table: torch.Tensor, sin_table: torch.Tensor, state: torch.Tensor,
                  kcache: torch.Tensor, vtcache: torch.Tensor) -> None:
    """mRoPE + cache append for the P rows of a pool step's fused q/k/v product (in place; fo1_pool_qkv_post_bf16)."""
    _chk(qkv, "qkv"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    p, ld, P, _ = _rows(qkv, "qkv")
    assert kcache.dim() == 3 and state.dtype == torch.int32 and state.is_contiguous() and state.shape[0] >= P
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    rc = _L.load().fo1_pool_qkv_post_bf16(p, ld, P, n_q, n_kv, head_dim, cos_table.data_ptr(), sin_table.data_ptr(), state.data_ptr(), kcache.data_ptr(),
                                          kcache.stride(0), pv, ldv, _stream())
    _L.check(rc, "fo1_pool_qkv_post_bf16")


def _need_ab(what: str) -> None:
    """Entry points of include/fo1_ab.h (measured no-gain kernel forms, instruments) exist in the test / bench build only."""
    if not _L.ab_build():
        raise _L.Fo1Error(f"{what} is part of include/fo1_ab.h: call it inside `with vlm_fo1_amd.lib.use_ab():` (or start the process with FO1_AB=1)")


def mfma_clock_probe(operands: int = 1, iters: int = 2000, workgroups: int = 256) -> dict:
    """Sustained clock / rate of a register-resident dense bf16 MFMA loop on this box (fo1_mfma_clock_probe, csrc/probe.hip):
    {clock_ghz (median over workgroups), tflops, us}.  operands 1 = pseudo-random bf16, 0 = zeros."""
    _need_ab("fo1_mfma_clock_probe")
    out = torch.zeros(workgroups, 2, dtype=torch.int64, device="cuda")
    sink = torch.zeros(1, dtype=torch.float32, device="cuda")
    for _ in range(2):          # the second launch is the measurement (the first ramps the clocks)
        _L.check(_L.load().fo1_mfma_clock_probe(int(operands), int(iters), int(workgroups), out.data_ptr(), sink.data_ptr(), _stream()), "fo1_mfma_clock_probe")
    torch.cuda.current_stream().synchronize()
    o = out.cpu().double()
    ghz = (o[:, 0] / (o[:, 1] * 10.0)).median().item()
    us = (o[:, 1] / 100.0).median().item()
    return dict(clock_ghz=round(ghz, 3), us=round(us, 1), tflops=round(workgroups * 8.0 * iters * 32.0 * 32768.0 / (us * 1e-6) / 1e12, 1))


def gemm_partials(a: torch.Tensor, w: torch.Tensor, splits: int, part: torch.Tensor) -> int:
    """Split-K planes of a @ w.T into part (fp32, >= splits * M * N elements): -> the effective number of planes [z, M, N] written
    (fo1_gemm_bf16_partials; no epilogue, no reduce — consumers: splitk_residual_rmsnorm, pool_qkv_post_partials)."""
    _chk(a, "a"); _chk(w, "w")
    pa, lda, M, K = _rows(a, "a")
    pw, ldw, N, Kw = _rows(w, "w")
    assert K == Kw and part.dtype == torch.float32 and part.is_contiguous() and part.numel() >= splits * M * N
    eff = ctypes.c_int(0)
    rc = _L.load().fo1_gemm_bf16_partials(pa, lda, pw, ldw, M, N, K, int(splits), part.data_ptr(), ctypes.byref(eff), _stream())
    _L.check(rc, "fo1_gemm_bf16_partials")
    return eff.value


def splitk_residual_rmsnorm(part: torch.Tensor, splits: int, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float, x_out: torch.Tensor,
                            xn_out: torch.Tensor, bias: Optional[torch.Tensor] = None) -> None:
    """x_out = bf16(bf16(sum_z part[z] (+ bias)) + residual); xn_out = RMSNorm(x_out) * norm_weight — one launch (fo1_splitk_residual_rmsnorm_bf16).
    x_out may alias residual (each element is read and written by the same thread)."""
    _chk(residual, "residual"); _chk(x_out, "x_out"); _chk(xn_out, "xn_out")
    pr, ldr, M, N = _rows(residual, "residual")
    px, ldx, _, _ = _rows(x_out, "x_out")
    pn, ldn, _, _ = _rows(xn_out, "xn_out")
    assert part.dtype == torch.float32 and part.numel() >= splits * M * N and norm_weight.numel() == N
    rc = _L.load().fo1_splitk_residual_rmsnorm_bf16(part.data_ptr(), int(splits), M, N, bias.data_ptr() if bias is not None else None, pr, ldr, px, ldx,
                                                    norm_weight.data_ptr(), float(eps), pn, ldn, _stream())
    _L.check(rc, "fo1_splitk_residual_rmsnorm_bf16")


def tile_weight(w: torch.Tensor) -> torch.Tensor:
    """The [N / 128][K / 64][128][64] copy of a row-major weight [N, K] that fo1_gemm_bf16_wtiled streams (N % 128 == 0, K % 64 == 0)."""
    N, K = w.shape
    assert N % 128 == 0 and K % 64 == 0 and w.dtype == torch.bfloat16
    return w.view(N // 128, 128, K // 64, 64).permute(0, 2, 1, 3).contiguous()


def gemm_wtiled(a: torch.Tensor, w_tiled: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """a @ W.T (+ epilogue) with W given as ops.tile_weight(W): the weight-streaming form for <= 128 rows (fo1_gemm_bf16_wtiled)."""
    _need_ab("fo1_gemm_bf16_wtiled")
    _chk(a, "a"); _chk(w_tiled, "w_tiled")
    pa, lda, M, K = _rows(a, "a")
    assert w_tiled.dim() == 4 and w_tiled.shape[2:] == (128, 64) and w_tiled.is_contiguous() and w_tiled.shape[1] * 64 == K
    N = w_tiled.shape[0] * 128
    out = torch.empty(M, N // 2 if act == ACT_SWIGLU16 else N, dtype=torch.bfloat16, device=a.device)
    po, ldc, _, _ = _rows(out, "out")
    pr, ldr = (None, 0)
    if residual is not None:
        _chk(residual, "residual")
        pr, ldr, _, _ = _rows(residual, "residual")
    rc = _L.load().fo1_gemm_bf16_wtiled(pa, lda, w_tiled.data_ptr(), bias.data_ptr() if bias is not None else None, pr, ldr, po, ldc, M, N, K, int(act), _stream())
    _L.check(rc, "fo1_gemm_bf16_wtiled")
    return out


def splitk_swiglu(part: torch.Tensor, splits: int, M: int, N: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out [M, N / 2] = SwiGLU over the split-K planes of a product against a 16-row interleaved gate/up weight (fo1_splitk_swiglu_bf16)."""
    _need_ab("fo1_splitk_swiglu_bf16")
    assert part.dtype == torch.float32 and part.numel() >= splits * M * N and part.is_cuda
    if out is None:
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=part.device)
    po, ldo, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_splitk_swiglu_bf16(part.data_ptr(), int(splits), M, N, po, ldo, _stream()), "fo1_splitk_swiglu_bf16")
    return out


def pool_qkv_post_partials(part: torch.Tensor, splits: int, bias: Optional[torch.Tensor], q_out: torch.Tensor, n_q: int, n_kv: int, head_dim: int,
                           cos_table: torch.Tensor, sin_table: torch.Tensor, state: torch.Tensor, kcache: torch.Tensor, vtcache: torch.Tensor) -> None:
    """pool_qkv_post fed by the split-K planes of the q/k/v projection: rotated q rows -> q_out [P, >= n_q * head_dim], K / V^T -> caches."""
    _chk(q_out, "q_out"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    p, ld, P, _ = _rows(q_out, "q_out")
    assert kcache.dim() == 3 and state.dtype == torch.int32 and state.is_contiguous() and state.shape[0] >= P
    assert part.dtype == torch.float32 and part.numel() >= splits * P * (n_q + 2 * n_kv) * head_dim
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    rc = _L.load().fo1_pool_qkv_post_partials_bf16(part.data_ptr(), int(splits), bias.data_ptr() if bias is not None else None, p, ld, P, n_q, n_kv, head_dim,
                                                   cos_table.data_ptr(), sin_table.data_ptr(), state.data_ptr(), kcache.data_ptr(), kcache.stride(0), pv, ldv,
                                                   _stream())
    _L.check(rc, "fo1_pool_qkv_post_partials_bf16")


def decode_argmax_accept(logits: Optional[torch.Tensor], first_tokens: Optional[torch.Tensor], state: torch.Tensor, plan: torch.Tensor,
                         ids_out: torch.Tensor, stop_ids: torch.Tensor, done: torch.Tensor, per_sequence_sets: bool = False) -> None:
    """per_sequence_sets: stop_ids is a table int32 [sets, 17] = {count, ids[16]} and state[b, 6] names sequence b's row (decode pool)."""
    B = state.shape[0]
    assert state.dtype == plan.dtype == ids_out.dtype == done.dtype == torch.int32 and ids_out.is_contiguous() and plan.is_contiguous()
    sc = _workspace("argmax_rows", state.device, 2 * 128 * B * 4)
    n_stop = int(stop_ids.numel()) if stop_ids is not None else 0
    if per_sequence_sets:
        assert stop_ids is not None and stop_ids.dim() == 2 and stop_ids.shape[1] == 17 and stop_ids.is_contiguous() and stop_ids.dtype == torch.int32
        n_stop = -1
    if logits is not None:
        _chk(logits, "logits")
        pl, ldl, Bl, V = _rows(logits, "logits")
        assert Bl == B
    else:
        pl, ldl, V = None, 0, 0
        assert first_tokens is not None and first_tokens.dtype == torch.int32 and first_tokens.numel() == B
    rc = _L.load().fo1_decode_argmax_accept(pl, ldl, V, B, first_tokens.data_ptr() if first_tokens is not None else None, state.data_ptr(),
                                            plan.data_ptr(), ids_out.data_ptr(), ids_out.shape[1], stop_ids.data_ptr() if n_stop else None, n_stop,
                                            done.data_ptr(), sc.data_ptr(), _stream())
    _L.check(rc, "fo1_decode_argmax_accept")


def kv_relocate(ksrc: torch.Tensor, kdst: torch.Tensor, vsrc: torch.Tensor, vdst: torch.Tensor, seqs: torch.Tensor, max_len: int) -> None:
    """k*: [layers, n_kv, rows, 128]; v*: [layers, n_kv*128, rows]; seqs int32 [B, 4] = (src0, dst0, len, 0) on the device."""
    for t in (ksrc, kdst, vsrc, vdst):
        _chk(t, "cache")
    assert ksrc.dim() == 4 and vsrc.dim() == 3 and seqs.dtype == torch.int32 and seqs.is_contiguous()
    rc = _L.load().fo1_kv_relocate(ksrc.data_ptr(), kdst.data_ptr(), ksrc.stride(0), ksrc.stride(1), kdst.stride(0), kdst.stride(1),
                                   vsrc.data_ptr(), vdst.data_ptr(), vsrc.stride(0), vsrc.stride(1), vdst.stride(0), vdst.stride(1),
                                   seqs.data_ptr(), seqs.shape[0], int(max_len), ksrc.shape[1], ksrc.shape[0], _stream())
    _L.check(rc, "fo1_kv_relocate")


def decode_qkv_post(qkv_row: torch.Tensor, n_q_heads: int, n_kv_heads: int, head_dim: int, cos_table: torch.Tensor,
                    sin_table: torch.Tensor, state: torch.Tensor, kcache: torch.Tensor, vtcache: torch.Tensor) -> None:
    _chk(qkv_row, "qkv_row"); _chk(kcache, "kcache"); _chk(vtcache, "vtcache")
    assert qkv_row.shape[0] == 1 and qkv_row.stride(1) == 1 and kcache.dim() == 3 and state.dtype == torch.int32
    pv, ldv, _, _ = _rows(vtcache, "vtcache")
    rc = _L.load().fo1_decode_qkv_post_bf16(qkv_row.data_ptr(), n_q_heads, n_kv_heads, head_dim, cos_table.data_ptr(), sin_table.data_ptr(),
                                            state.data_ptr(), kcache.data_ptr(), kcache.stride(0), pv, ldv, _stream())
    _L.check(rc, "fo1_decode_qkv_post_bf16")


def decode_advance(state: torch.Tensor) -> None:
    assert state.dtype == torch.int32 and state.numel() >= 8 and state.is_contiguous()
    _L.check(_L.load().fo1_decode_advance(state.data_ptr(), _stream()), "fo1_decode_advance")


ATTN32_MAX_ROW_BYTES = 8192     # widest K row pitch of the engine (bytes): see pick_q_block


def _check_attn32_extent(q_block: int, rows: int, row_bytes: int) -> None:
    """fo1_attention_bf16 with q_block 128 / 256 builds 32-bit byte offsets from the K base: fail loudly instead of reading zeros."""
    if q_block >= 128 and rows * row_bytes >= 2 ** 32:
        raise ValueError(f"attention: q_block {q_block} addresses K rows with 32-bit byte offsets; {rows} rows x {row_bytes} B exceed 4 GiB — build the items with q_block 64")


def pick_q_block(segments: Sequence[Sequence[int]], n_heads: int, head_dim: int = 0, n_kv_heads: Optional[int] = None) -> int:
    """Query block of an attention work list.
    64 (/ 32 / 16) = the 16x16-MFMA kernel, 4 / 2 / 1 waves x 16 queries per workgroup — smaller blocks re-stage every K/V tile once
    per block and LOSE (LLM prefill L=515: 64 -> 1.9 ms, auto 16/32 -> 4.7 ms per image), so 64 is the floor.
    128 / 256 = the 32x32-MFMA kernel (attn_fwd32_kernel, head dim 80 / 128, round 5): 8 waves x 32 queries per workgroup —
    128 queries x the TWO query heads of one KV head when the model is grouped-query (LLM prefill: both heads share the staged
    K / V^T tile), 256 queries of one head otherwise (ViT full attention).  Short segments (ViT windows of 64 tokens, DaViT's 144-token
    windows at head dim 32, one-token extends) stay on the 64-query kernel.  FO1_ATTN32=0 turns the new form off (A/B)."""
    if head_dim in (80, 128) and os.environ.get("FO1_ATTN32", "1") != "0" and len(segments):
        # the 32x32 form addresses K / V^T rows through buffer descriptors with 32-bit byte offsets from the operand's base: the
        # last key row must stay below 2^32 bytes at the WIDEST row pitch the engine uses (head-major ViT q/k/v: 16 heads x 256 elements =
        # 8 KB per token) — 524k rows; past that the 16x16 kernel (64-bit addresses) takes the launch
        if max(int(e) for _, e, *_ in segments) * ATTN32_MAX_ROW_BYTES >= 2 ** 32:
            return 64
        longest = max(int(e) - int(s) for s, e, *_ in segments)
        group = n_heads // (n_kv_heads or n_heads)
        if head_dim == 128 and group % 2 == 0 and longest > 64:
            return 128
        if longest >= 192:
            return 256
    return 64


def order_items(rows, block: int, causal: bool, prefix=None):
    """Work-list order of the 32x32 attention form (q_block >= 128): descending cost = key tiles the item walks (its own range — up to its
    last query when causal — plus its second range).  The kernel's grid walks the items slowest, so this is longest-processing-time-first
    over the launch: the few-tile blocks fill the tail instead of a 48-tile block starting last (hires LLM: items of 2..48 tiles).
    A pure permutation of the work: every query's arithmetic is unchanged.  -> index permutation."""
    if block < 128:
        return list(range(len(rows)))
    def cost(i):
        q0, q1, k0, k1 = rows[i]
        own = (min(k1, q1) if causal else k1) - k0
        pre = 0 if prefix is None else max(0, prefix[i][1] - prefix[i][0])
        return -((own + 63) // 64 + (pre + 63) // 64)
    return sorted(range(len(rows)), key=cost)


def make_items(segments: Sequence[Sequence[int]], device, causal: bool = False, block: int = 64) -> torch.Tensor:
    """Host-side work list for fo1_attention_bf16: split every segment [start, end) into query blocks
    of <= block.  Index bookkeeping belongs on the host.  The block size rides along as
    attribute `q_block` of the returned tensor."""
    rows = []
    for s, e in segments:
        for q0 in range(s, e, block):
            rows.append((q0, min(q0 + block, e), s, e))
    rows = [rows[i] for i in order_items(rows, block, causal)]
    t = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4).to(device)
    t.q_block = block
    return t


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, items: torch.Tensor, n_q_heads: int, n_kv_heads: int,
              head_dim: int, scale: float, causal: bool, out: Optional[torch.Tensor] = None,
              flops: float = 0.0, qk_head_stride: Optional[int] = None) -> torch.Tensor:
    """q: [L, >= n_q_heads*head_dim] view (heads contiguous), k: [Lk, ...] view, vt: [n_kv_heads*head_dim, ld] (V^T).
    qk_head_stride: elements between consecutive heads of q and of k when they are not packed (the head-major q/k/v layout of
    qkv_proj_rope mode 1: 256).  Returns out [L, n_q_heads*head_dim]."""
    _chk(q, "q"); _chk(k, "k"); _chk(vt, "vt")
    assert items.dtype == torch.int32 and items.is_contiguous() and items.device == q.device
    pq, ldq, L, _ = _rows(q, "q")
    pk, ldk, _, _ = _rows(k, "k")
    pv, ldv, _, _ = _rows(vt, "vt")
    if out is None:
        out = torch.empty(L, n_q_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    po, ldo, _, _ = _rows(out, "out")
    hs = head_dim if qk_head_stride is None else int(qk_head_stride)
    _check_attn32_extent(getattr(items, "q_block", 64), k.shape[0], ldk * 2)
    rc = _L.load().fo1_attention_bf16(pq, ldq, hs, pk, ldk, hs, pv, ldv, po, ldo, head_dim,
                                      items.data_ptr(), items.shape[0], getattr(items, "q_block", 64), n_q_heads, n_kv_heads,
                                      head_dim, float(scale), 1 if causal else 0, None, float(flops), _stream())
    _L.check(rc, "fo1_attention_bf16")
    return out


def single_tile_items(segments: Sequence[Sequence[int]], head_dim: int) -> bool:
    """True when the work list of these segments can go to fo1_attention_windows_bf16: head dim 80 and every segment at most 64 tokens (make_items
    with block 64 then lists one item per segment).  FO1_WINATTN=0 turns it off (A/B)."""
    if head_dim != 80 or os.environ.get("FO1_WINATTN", "1") == "0" or not len(segments):
        return False
    return all(0 < int(e) - int(s) <= 64 for s, e, *_ in segments)


def attention_windows(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, items: torch.Tensor, n_heads: int, head_dim: int, scale: float,
                      flops: float = 0.0, qk_head_stride: Optional[int] = None) -> torch.Tensor:
    """ops.attention (non-causal, q_block 64) for a work list of single-tile items (items.single_tile, see single_tile_items): the pipelined
    fo1_attention_windows_bf16, the same bits."""
    _chk(q, "q"); _chk(k, "k"); _chk(vt, "vt")
    assert items.dtype == torch.int32 and items.is_contiguous() and items.device == q.device and getattr(items, "single_tile", False)
    pq, ldq, L, _ = _rows(q, "q")
    pk, ldk, _, _ = _rows(k, "k")
    pv, ldv, _, _ = _rows(vt, "vt")
    out = torch.empty(L, n_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    hs = head_dim if qk_head_stride is None else int(qk_head_stride)
    rc = _L.load().fo1_attention_windows_bf16(pq, ldq, hs, pk, ldk, hs, pv, ldv, out.data_ptr(), out.stride(0), head_dim, L, items.data_ptr(), items.shape[0],
                                              n_heads, n_heads, head_dim, float(scale), float(flops), _stream())
    _L.check(rc, "fo1_attention_windows_bf16")
    return out


# ---- DaViT / SimpleFPN / splice helpers -----------------------------------------------------
# Every spatial op takes `batch`: that many same-size images stacked along the row dimension ([batch*H*W, C]).
def dwconv3x3_res(x: torch.Tensor, w9c: torch.Tensor, bias: torch.Tensor, H: int, W: int, batch: int = 1) -> torch.Tensor:
    """x [batch*H*W, C] token-major -> x + dwconv3x3(x) (+bias); w9c is the [9, C] tap-major weight."""
    _chk(x, "x"); _chk(w9c, "w9c"); _chk(bias, "bias")
    assert x.is_contiguous() and x.shape[0] == batch * H * W and w9c.shape == (9, x.shape[1]) and w9c.is_contiguous()
    y = torch.empty_like(x)
    _L.check(_L.load().fo1_dwconv3x3_bf16(x.data_ptr(), w9c.data_ptr(), bias.data_ptr(), y.data_ptr(), H, W, x.shape[1], batch, _stream()),
             "fo1_dwconv3x3_bf16")
    return y


def dwconv3x3_res_ln(x: torch.Tensor, w9c: torch.Tensor, bias: torch.Tensor, H: int, W: int, ln_w: torch.Tensor, ln_b: torch.Tensor,
                     eps: float, batch: int = 1):
    """-> (y = x + dwconv3x3(x) + bias, LayerNorm(y)) in one launch (fo1_dwconv3x3_ln_bf16), bit-identical to
    dwconv3x3_res followed by layernorm."""
    _chk(x, "x"); _chk(w9c, "w9c"); _chk(bias, "bias"); _chk(ln_w, "ln_w"); _chk(ln_b, "ln_b")
    assert x.is_contiguous() and x.shape[0] == batch * H * W and w9c.shape == (9, x.shape[1]) and w9c.is_contiguous()
    y, h = torch.empty_like(x), torch.empty_like(x)
    _L.check(_L.load().fo1_dwconv3x3_ln_bf16(x.data_ptr(), w9c.data_ptr(), bias.data_ptr(), y.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                             float(eps), h.data_ptr(), H, W, x.shape[1], batch, _stream()), "fo1_dwconv3x3_ln_bf16")
    return y, h


def im2col(x: torch.Tensor, H: int, W: int, KH: int, KW: int, stride: int, pad: int, ld: Optional[int] = None, batch: int = 1):
    """x [batch*H*W, C] -> (col [batch*Ho*Wo, ld>=KH*KW*C] (pad columns zero), Ho, Wo)."""
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == batch * H * W
    C = x.shape[1]
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    K = KH * KW * C
    ld = ld or K
    col = torch.zeros(batch * Ho * Wo, ld, dtype=torch.bfloat16, device=x.device) if ld != K else \
        torch.empty(batch * Ho * Wo, ld, dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_im2col_bf16(x.data_ptr(), col.data_ptr(), H, W, C, KH, KW, stride, pad, ld, batch, _stream()), "fo1_im2col_bf16")
    return col, Ho, Wo


def window_partition(x: torch.Tensor, H: int, W: int, ws: int, batch: int = 1) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == batch * H * W
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    xw = torch.empty(batch * nW * ws * ws, x.shape[1], dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_window_partition_bf16(x.data_ptr(), xw.data_ptr(), H, W, x.shape[1], ws, batch, _stream()),
             "fo1_window_partition_bf16")
    return xw


def window_reverse_add(yw: torch.Tensor, shortcut: torch.Tensor, H: int, W: int, ws: int, batch: int = 1) -> torch.Tensor:
    _chk(yw, "yw"); _chk(shortcut, "shortcut")
    assert yw.is_contiguous() and shortcut.is_contiguous() and shortcut.shape[0] == batch * H * W
    y = torch.empty_like(shortcut)
    _L.check(_L.load().fo1_window_reverse_add_bf16(yw.data_ptr(), shortcut.data_ptr(), y.data_ptr(), H, W, shortcut.shape[1], ws, batch,
                                                   _stream()), "fo1_window_reverse_add_bf16")
    return y


WINDOW_ATTENTION_MAX_TOKENS = 160      # fo1_window_attention_bf16: tokens per window the head-dim-32 kernel is built for


def window_attention(qkv: torch.Tensor, C: int, n_heads: int, window_tokens: int, scale: float) -> torch.Tensor:
    """qkv [n_windows * window_tokens, 3C] (the q/k/v GEMM's rows of window-partitioned tokens) -> softmax(q k^T * scale) v per window and
    head, [rows, C] (fo1_window_attention_bf16: head dim 32, DaViT's WindowAttention modeling_davit.py:225-282)."""
    _chk(qkv, "qkv")
    p, ld, n, _ = _rows(qkv, "qkv")
    assert C == n_heads * 32 and n % window_tokens == 0 and window_tokens <= WINDOW_ATTENTION_MAX_TOKENS
    out = torch.empty(n, C, dtype=torch.bfloat16, device=qkv.device)
    _L.check(_L.load().fo1_window_attention_bf16(p, ld, C, n_heads, window_tokens, n // window_tokens, out.data_ptr(), C, float(scale), _stream()),
             "fo1_window_attention_bf16")
    return out


WINDOW_ATTENTION_MAP_WINDOW = 12       # fo1_window_attention_map_bf16: the window side its token -> pixel arithmetic is built for


def window_attention_map(qkv: torch.Tensor, C: int, n_heads: int, window: int, H: int, W: int, batch: int, pad_row: torch.Tensor, scale: float) -> torch.Tensor:
    """Window attention on UN-partitioned rows: qkv [batch * H * W, 3C] = the q/k/v projection of the images' pixels in raster order; the windows'
    tokens are found by arithmetic, tokens outside the image read `pad_row` (the layer's bf16 q/k/v bias = the projection of the reference's zero
    padding, modeling_davit.py:248-251).  -> [batch * H * W, C] in pixel order (fo1_window_attention_map_bf16)."""
    _chk(qkv, "qkv"); _chk(pad_row, "pad_row")
    p, ld, n, _ = _rows(qkv, "qkv")
    assert C == n_heads * 32 and n == batch * H * W and window == WINDOW_ATTENTION_MAP_WINDOW and pad_row.numel() == 3 * C and pad_row.is_contiguous()
    out = torch.empty(n, C, dtype=torch.bfloat16, device=qkv.device)
    _L.check(_L.load().fo1_window_attention_map_bf16(p, ld, C, n_heads, window, H, W, batch, pad_row.data_ptr(), out.data_ptr(), C, float(scale), _stream()),
             "fo1_window_attention_map_bf16")
    return out


def window_attention_map_var(qkv: torch.Tensor, C: int, n_heads: int, window: int, sg: "ImgSegs", pad_row: torch.Tensor, scale: float) -> torch.Tensor:
    """window_attention_map for images of different sizes (sg = the window-partition geometry table: pixels row0, H, W, -, windows down / across)."""
    _chk(qkv, "qkv"); _chk(pad_row, "pad_row")
    p, ld, n, _ = _rows(qkv, "qkv")
    assert C == n_heads * 32 and n == sg.total_in and window == WINDOW_ATTENTION_MAP_WINDOW and pad_row.numel() == 3 * C and pad_row.is_contiguous()
    out = torch.empty(n, C, dtype=torch.bfloat16, device=qkv.device)
    _L.check(_L.load().fo1_window_attention_map_var_bf16(p, ld, C, n_heads, window, sg.ptr, sg.n, sg.max_out // (window * window), sg.total_in,
                                                         pad_row.data_ptr(), out.data_ptr(), C, float(scale), _stream()), "fo1_window_attention_map_var_bf16")
    return out


def channel_attention(qkv: torch.Tensor, C: int, batch: int = 1) -> torch.Tensor:
    """qkv [batch*N, 3C]: per image, per 32-channel group attention over the image's own N tokens."""
    _chk(qkv, "qkv")
    p, ld, NB, _ = _rows(qkv, "qkv")
    assert NB % batch == 0 and (batch == 1 or qkv.is_contiguous() or qkv.stride(0) == ld)
    N = NB // batch
    need = _L.load().fo1_channel_attention_workspace_bytes(N, C, batch)
    ws = _workspace("channel_attention", qkv.device, need)
    out = torch.empty(NB, C, dtype=torch.bfloat16, device=qkv.device)
    _L.check(_L.load().fo1_channel_attention_bf16(p, ld, N, C, out.data_ptr(), C, batch, ws.data_ptr(), ws.numel(), _stream()),
             "fo1_channel_attention_bf16")
    return out


def pixel_shuffle2(src: torch.Tensor, H: int, W: int, Co: int, batch: int = 1) -> torch.Tensor:
    _chk(src, "src")
    assert src.is_contiguous() and src.shape == (batch * H * W, 4 * Co)
    dst = torch.empty(batch * 4 * H * W, Co, dtype=torch.bfloat16, device=src.device)
    _L.check(_L.load().fo1_pixel_shuffle2_bf16(src.data_ptr(), dst.data_ptr(), H, W, Co, batch, _stream()), "fo1_pixel_shuffle2_bf16")
    return dst


def maxpool2(x: torch.Tensor, H: int, W: int, batch: int = 1) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == batch * H * W
    y = torch.empty(batch * (H // 2) * (W // 2), x.shape[1], dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_maxpool2_bf16(x.data_ptr(), y.data_ptr(), H, W, x.shape[1], batch, _stream()), "fo1_maxpool2_bf16")
    return y


# ---- ragged image batches: images of different sizes packed row-wise (include/fo1.h fo1_img_seg) ---------------------------------
class ImgSegs:
    """Per-operator geometry table of a ragged batch: host rows [(in_row0, H, W, out_row0, Ho, Wo)] -> device int32 [n, 8] + the sizes
    the launch needs (largest image, totals).  Built once per batch signature (the tower plans cache them)."""

    def __init__(self, rows, device, max_in: int, total_in: int, max_out: int, total_out: int):
        t = torch.zeros(len(rows), 8, dtype=torch.int32)
        for i, r in enumerate(rows):
            t[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
        self.dev = t.to(device)
        self.n = len(rows)
        self.max_in, self.total_in, self.max_out, self.total_out = int(max_in), int(total_in), int(max_out), int(total_out)

    @property
    def ptr(self):
        return self.dev.data_ptr()


def dwconv3x3_res_ln_var(x: torch.Tensor, w9c: torch.Tensor, bias: torch.Tensor, sg: ImgSegs, ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float):
    _chk(x, "x"); _chk(w9c, "w9c"); _chk(bias, "bias"); _chk(ln_w, "ln_w"); _chk(ln_b, "ln_b")
    assert x.is_contiguous() and x.shape[0] == sg.total_in and w9c.shape == (9, x.shape[1]) and w9c.is_contiguous()
    y, h = torch.empty_like(x), torch.empty_like(x)
    _L.check(_L.load().fo1_dwconv3x3_ln_var_bf16(x.data_ptr(), w9c.data_ptr(), bias.data_ptr(), y.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                                 float(eps), h.data_ptr(), sg.ptr, sg.n, sg.max_in, sg.total_in, x.shape[1], _stream()),
             "fo1_dwconv3x3_ln_var_bf16")
    return y, h


def im2col_var(x: torch.Tensor, sg: ImgSegs, KH: int, KW: int, stride: int, pad: int, ld: Optional[int] = None) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == sg.total_in
    C = x.shape[1]
    K = KH * KW * C
    ld = ld or K
    col = torch.zeros(sg.total_out, ld, dtype=torch.bfloat16, device=x.device) if ld != K else \
        torch.empty(sg.total_out, ld, dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_im2col_var_bf16(x.data_ptr(), col.data_ptr(), sg.ptr, sg.n, sg.max_out, sg.total_out, C, KH, KW, stride, pad, ld, _stream()),
             "fo1_im2col_var_bf16")
    return col


def window_partition_var(x: torch.Tensor, sg: ImgSegs, ws: int) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == sg.total_in
    xw = torch.empty(sg.total_out, x.shape[1], dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_window_partition_var_bf16(x.data_ptr(), xw.data_ptr(), sg.ptr, sg.n, sg.max_out, sg.total_out, x.shape[1], ws, _stream()),
             "fo1_window_partition_var_bf16")
    return xw


def window_reverse_add_var(yw: torch.Tensor, shortcut: torch.Tensor, sg: ImgSegs, ws: int) -> torch.Tensor:
    _chk(yw, "yw"); _chk(shortcut, "shortcut")
    assert yw.is_contiguous() and shortcut.is_contiguous() and shortcut.shape[0] == sg.total_in and yw.shape[0] == sg.total_out
    y = torch.empty_like(shortcut)
    _L.check(_L.load().fo1_window_reverse_add_var_bf16(yw.data_ptr(), shortcut.data_ptr(), y.data_ptr(), sg.ptr, sg.n, sg.max_in, sg.total_in,
                                                       shortcut.shape[1], ws, _stream()), "fo1_window_reverse_add_var_bf16")
    return y


def channel_attention_var(qkv: torch.Tensor, C: int, sg: ImgSegs) -> torch.Tensor:
    """qkv [sum N_i, 3C]: per image, per 32-channel group attention over the image's own N_i tokens (sg rows: (row0, N_i))."""
    _chk(qkv, "qkv")
    p, ld, NB, _ = _rows(qkv, "qkv")
    assert NB == sg.total_in
    need = _L.load().fo1_channel_attention_var_workspace_bytes(sg.max_in, C, sg.n)
    ws = _workspace("channel_attention", qkv.device, need)
    out = torch.empty(NB, C, dtype=torch.bfloat16, device=qkv.device)
    _L.check(_L.load().fo1_channel_attention_var_bf16(p, ld, sg.ptr, sg.n, sg.max_in, sg.total_in, C, out.data_ptr(), C, ws.data_ptr(), ws.numel(),
                                                      _stream()), "fo1_channel_attention_var_bf16")
    return out


def pixel_shuffle2_var(src: torch.Tensor, sg: ImgSegs, Co: int) -> torch.Tensor:
    _chk(src, "src")
    assert src.is_contiguous() and src.shape == (sg.total_in, 4 * Co)
    dst = torch.empty(sg.total_out, Co, dtype=torch.bfloat16, device=src.device)
    _L.check(_L.load().fo1_pixel_shuffle2_var_bf16(src.data_ptr(), dst.data_ptr(), sg.ptr, sg.n, sg.max_in, sg.total_in, Co, _stream()),
             "fo1_pixel_shuffle2_var_bf16")
    return dst


def maxpool2_var(x: torch.Tensor, sg: ImgSegs) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == sg.total_in
    y = torch.empty(sg.total_out, x.shape[1], dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_maxpool2_var_bf16(x.data_ptr(), y.data_ptr(), sg.ptr, sg.n, sg.max_out, sg.total_out, x.shape[1], _stream()),
             "fo1_maxpool2_var_bf16")
    return y


def nchw_to_hwc8(img: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """img [3,H,W] or [B,3,H,W] bf16/fp32 -> [B*H*W, 8] bf16 (into `out` when given: the image's rows of a packed ragged batch)."""
    if img.device.type != "cuda":
        raise _L.Fo1Error("nchw_to_hwc8: expected a HIP device tensor")
    if img.dim() == 3:
        img = img.unsqueeze(0)
    assert img.dim() == 4 and img.shape[1] == 3 and img.is_contiguous() and img.dtype in (torch.bfloat16, torch.float32)
    B, _, H, W = img.shape
    if out is None:
        out = torch.empty(B * H * W, 8, dtype=torch.bfloat16, device=img.device)
    else:
        assert out.shape == (B * H * W, 8) and out.is_contiguous() and out.dtype == torch.bfloat16
    _L.check(_L.load().fo1_nchw_to_hwc8_bf16(img.data_ptr(), 1 if img.dtype == torch.float32 else 0, out.data_ptr(), H, W, B, _stream()),
             "fo1_nchw_to_hwc8_bf16")
    return out


def gather_rows(plan: torch.Tensor, D: int, t0: torch.Tensor, t1: Optional[torch.Tensor] = None, t2: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None):
    """out[r, :D] = table[plan[r,0]][plan[r,1], :D];  plan int32 [R,2] (kind, index) on device.  `out` may be
    wider than D (its other columns are left untouched)."""
    assert plan.dtype == torch.int32 and plan.is_contiguous()
    R = plan.shape[0]
    if out is None:
        out = torch.empty(R, D, dtype=torch.bfloat16, device=plan.device)
    po, ldo, Ro, Do = _rows(out, "out")
    assert Ro == R and Do >= D

    def pl(t):
        if t is None:
            return None, 0
        _chk(t, "table")
        assert t.dim() == 2 and t.stride(1) == 1 and t.shape[1] >= D
        return t.data_ptr(), t.stride(0)

    p0, l0 = pl(t0); p1, l1 = pl(t1); p2, l2 = pl(t2)
    _L.check(_L.load().fo1_gather_rows_bf16(p0, l0, p1, l1, p2, l2, plan.data_ptr(), po, ldo, R, D, _stream()),
             "fo1_gather_rows_bf16")
    return out


def gather_rows_into(plan: torch.Tensor, D: int, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    return gather_rows(plan, D, table, out=out)


def attention_strided(q: torch.Tensor, q_row0: int, k: torch.Tensor, vt: torch.Tensor, items: torch.Tensor, n_q_heads: int,
                      n_kv_heads: int, head_dim: int, scale: float, causal: bool, flops: float = 0.0,
                      out: Optional[torch.Tensor] = None, q_row_base: Optional[torch.Tensor] = None, n_items: Optional[int] = None,
                      prefix_ranges: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention against a KV cache.  q [Lq, n_q_heads*head_dim] holds absolute positions q_row0 .. q_row0+Lq (the
    items index absolute positions) — or, with `q_row_base` (device int32 scalar), positions *q_row_base + row so a
    captured graph can serve every decode step.  k is the cache [n_kv, Lmax, head_dim], vt the transposed V cache
    [n_kv*head_dim, Lmax]; `items` may be a raw int32 view (n_items rows of 4)."""
    _chk(q, "q"); _chk(k, "k"); _chk(vt, "vt")
    assert items.dtype == torch.int32 and items.is_contiguous()
    pq, ldq, Lq, _ = _rows(q, "q")
    assert k.dim() == 3 and k.stride(2) == 1
    pv, ldv, _, _ = _rows(vt, "vt")
    if out is None:
        out = torch.empty(Lq, n_q_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    po, ldo, _, _ = _rows(out, "out")
    if q_row_base is not None:
        q_row0 = 0
    _check_attn32_extent(getattr(items, "q_block", 64), k.shape[1], k.stride(1) * 2)
    if prefix_ranges is not None:      # items with a second (shared-prefix) key range: fo1_attention_prefix_bf16
        assert prefix_ranges.dtype == torch.int32 and prefix_ranges.is_contiguous() and prefix_ranges.shape == (items.shape[0], 2) and q_row_base is None
        rc = _L.load().fo1_attention_prefix_bf16(pq - q_row0 * ldq * 2, ldq, head_dim, k.data_ptr(), k.stride(1), k.stride(0), pv, ldv,
                                                 po - q_row0 * ldo * 2, ldo, head_dim, items.data_ptr(), prefix_ranges.data_ptr(), items.shape[0],
                                                 getattr(items, "q_block", 64), n_q_heads, n_kv_heads, head_dim, float(scale), 1 if causal else 0,
                                                 float(flops), _stream())
        _L.check(rc, "fo1_attention_prefix_bf16")
        return out
    rc = _L.load().fo1_attention_bf16(pq - q_row0 * ldq * 2, ldq, head_dim, k.data_ptr(), k.stride(1), k.stride(0), pv, ldv,
                                      po - q_row0 * ldo * 2, ldo, head_dim, items.data_ptr(),
                                      n_items if n_items is not None else items.shape[0], getattr(items, "q_block", 64),
                                      n_q_heads, n_kv_heads, head_dim, float(scale), 1 if causal else 0,
                                      q_row_base.data_ptr() if q_row_base is not None else None, float(flops), _stream())
    _L.check(rc, "fo1_attention_bf16")
    return out


# ---- image preprocessing (device side) -------------------------------------------------------------------------------------
def patchify_u8(image_hwc: torch.Tensor, lut: torch.Tensor, patch: int = 14, merge: int = 2) -> torch.Tensor:
    """uint8 [H, W, 3] (device) -> bf16 [S, 6*patch^2] normalised patches in merge-block order (fo1_patchify_u8_bf16)."""
    if image_hwc.dtype != torch.uint8 or not image_hwc.is_cuda or image_hwc.dim() != 3 or image_hwc.shape[2] != 3 or not image_hwc.is_contiguous():
        raise ValueError("patchify_u8: need a contiguous device uint8 [H, W, 3] tensor")
    _chk(lut, "lut")
    assert lut.shape == (3, 256) and lut.is_contiguous()
    H, W, _ = image_hwc.shape
    out = torch.empty((H // patch) * (W // patch), 6 * patch * patch, dtype=torch.bfloat16, device=image_hwc.device)
    _L.check(_L.load().fo1_patchify_u8_bf16(image_hwc.data_ptr(), H, W, lut.data_ptr(), out.data_ptr(), out.stride(0), patch, merge,
                                            _stream()), "fo1_patchify_u8_bf16")
    return out


def normalize_u8(image_hwc: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """uint8 [H, W, 3] (device) -> bf16 [3, H, W] normalised, channels first (fo1_normalize_u8_bf16)."""
    if image_hwc.dtype != torch.uint8 or not image_hwc.is_cuda or image_hwc.dim() != 3 or image_hwc.shape[2] != 3 or not image_hwc.is_contiguous():
        raise ValueError("normalize_u8: need a contiguous device uint8 [H, W, 3] tensor")
    _chk(lut, "lut")
    assert lut.shape == (3, 256) and lut.is_contiguous()
    H, W, _ = image_hwc.shape
    out = torch.empty(3, H, W, dtype=torch.bfloat16, device=image_hwc.device)
    _L.check(_L.load().fo1_normalize_u8_bf16(image_hwc.data_ptr(), H, W, lut.data_ptr(), out.data_ptr(), _stream()),
             "fo1_normalize_u8_bf16")
    return out


# ---- multi-scale deformable attention (msda.hip; UPN proposal detector) ----------------------------------
def ms_deform_attn(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_locations: torch.Tensor,
                   attention_weights: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """value [N, S, M, D] (fp32 | fp64 | bf16), spatial_shapes int64 [L, 2] and level_start_index int64 [L] on the device,
    sampling_locations [N, Lq, M, L, P, 2], attention_weights [N, Lq, M, L, P] (value's dtype; fp32 when value is bf16)
    -> [N, Lq, M*D] in value's dtype (fo1_ms_deform_attn_forward)."""
    for t, name in ((value, "value"), (spatial_shapes, "spatial_shapes"), (level_start_index, "level_start_index"),
                    (sampling_locations, "sampling_locations"), (attention_weights, "attention_weights")):
        if not t.is_cuda:
            raise RuntimeError(f"ms_deform_attn: {name} must live on the GPU (no CPU path exists)")
        if not t.is_contiguous():
            raise ValueError(f"ms_deform_attn: {name} must be contiguous")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise TypeError("ms_deform_attn: spatial_shapes / level_start_index must be int64 (as the reference passes them)")
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_locations.shape
    if M2 != M or two != 2 or tuple(attention_weights.shape) != (N, Lq, M, L, P) or tuple(spatial_shapes.shape) != (L, 2) or level_start_index.numel() != L:
        raise ValueError("ms_deform_attn: inconsistent shapes")
    if value.dtype == torch.float32:
        dt, lt = 0, torch.float32
    elif value.dtype == torch.float64:
        dt, lt = 1, torch.float64
    elif value.dtype == torch.bfloat16:
        dt, lt = 2, torch.float32
    else:
        raise TypeError(f"ms_deform_attn: value dtype {value.dtype} not built (float32, float64, bfloat16)")
    if sampling_locations.dtype != lt or attention_weights.dtype != lt:
        raise TypeError(f"ms_deform_attn: sampling_locations / attention_weights must be {lt} for {value.dtype} values")
    if out is None:
        out = torch.empty(N, Lq, M * D, dtype=value.dtype, device=value.device)
    rc = _L.load().fo1_ms_deform_attn_forward(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_locations.data_ptr(),
                                              attention_weights.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), dt, _stream())
    _L.check(rc, "fo1_ms_deform_attn_forward")
    return out


def msda_fused(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, offsets_logits: torch.Tensor,
               reference_points: torch.Tensor, n_heads: int, n_points: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused MSDeformAttn core (fo1_msda_fused_bf16): value bf16 [N, S, C]; offsets_logits fp32 [N, Lq, M*L*P*3]
    ([sampling_offsets | attention_weights] rows of one GEMM); reference_points fp32 [N, Lq, L or 1, 2 | 4] -> bf16 [N, Lq, C]."""
    _chk(value, "value")
    for t, name in ((offsets_logits, "offsets_logits"), (reference_points, "reference_points")):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"msda_fused: {name} must be a contiguous fp32 GPU tensor")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64 or not spatial_shapes.is_cuda or not level_start_index.is_cuda:
        raise TypeError("msda_fused: spatial_shapes / level_start_index must be int64 GPU tensors")
    N, S, C = value.shape
    _, Lq, RL, RD = reference_points.shape
    L = spatial_shapes.shape[0]
    M, P = n_heads, n_points
    D = C // M
    if not value.is_contiguous() or tuple(offsets_logits.shape) != (N, Lq, M * L * P * 3) or RL not in (1, L):
        raise ValueError("msda_fused: inconsistent shapes")
    if out is None:
        out = torch.empty(N, Lq, C, dtype=torch.bfloat16, device=value.device)
    rc = _L.load().fo1_msda_fused_bf16(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), offsets_logits.data_ptr(),
                                       reference_points.data_ptr(), RL, RD, N, S, M, D, L, Lq, P, out.data_ptr(), _stream())
    _L.check(rc, "fo1_msda_fused_bf16")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = bf16(a + b) over [M, D] rows (fo1_add_bf16)."""
    _chk(a, "a"); _chk(b, "b")
    pa, lda, M, D = _rows(a, "a")
    pb, ldb, Mb, Db = _rows(b, "b")
    assert (M, D) == (Mb, Db)
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=a.device)
    po, ldo, _, _ = _rows(out, "out")
    _L.check(_L.load().fo1_add_bf16(pa, lda, pb, ldb, po, ldo, M, D, _stream()), "fo1_add_bf16")
    return out


# ---- UPN query selection / decoder helpers (upn_ops.hip) -------------------------------------------------------------------
def _f32(t: torch.Tensor, name: str):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name} must be a 2-d fp32 GPU tensor with unit column stride")
    return t.data_ptr(), t.stride(0), t.shape[0], t.shape[1]


def sine_embed(ref: torch.Tensor, dims: int = 4) -> torch.Tensor:
    """fp32 [n, >= dims] (x, y[, w, h]) -> bf16 [n, dims*128] (fo1_sine_embed_bf16)."""
    p, ld, n, _ = _f32(ref, "ref")
    out = torch.empty(n, dims * 128, dtype=torch.bfloat16, device=ref.device)
    _L.check(_L.load().fo1_sine_embed_bf16(p, ld, n, dims, out.data_ptr(), out.stride(0), _stream()), "fo1_sine_embed_bf16")
    return out


def box_refine(delta: torch.Tensor, ref: torch.Tensor, mode: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mode 0: sigmoid(delta[:, :4] + inverse_sigmoid(ref)); mode 1: delta[:, :4] + ref; mode 2: sigmoid(delta[:, :4] + ref).
    fp32 [n, 4] (fo1_box_refine_f32)."""
    pd, ldd, n, _ = _f32(delta, "delta")
    pr, ldr, nr, _ = _f32(ref, "ref")
    assert n == nr
    if out is None:
        out = torch.empty(n, 4, dtype=torch.float32, device=delta.device)
    po, ldo, _, _ = _f32(out, "out")
    _L.check(_L.load().fo1_box_refine_f32(pd, ldd, pr, ldr, po, ldo, n, mode, _stream()), "fo1_box_refine_f32")
    return out


def mask_rows(x: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    _chk(x, "x")
    px, ldx, M, D = _rows(x, "x")
    assert keep.dtype == torch.uint8 and keep.is_cuda and keep.numel() == M and keep.is_contiguous()
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_mask_rows_bf16(px, ldx, keep.data_ptr(), y.data_ptr(), y.stride(0), M, D, _stream()), "fo1_mask_rows_bf16")
    return y


def topk_desc(scores: torch.Tensor, k: int, stride: int = 1, n: Optional[int] = None):
    """scores fp32 (flat; element i at i*stride) -> (indices int32 [k], values fp32 [k]) in descending order (fo1_topk_desc_f32)."""
    if not scores.is_cuda or scores.dtype != torch.float32 or not scores.is_contiguous():
        raise TypeError("topk_desc: scores must be a contiguous fp32 GPU tensor")
    n = scores.numel() // stride if n is None else n
    ws = _workspace("topk", scores.device, _L.load().fo1_topk_workspace_bytes(n))
    idx = torch.empty(k, dtype=torch.int32, device=scores.device)
    val = torch.empty(k, dtype=torch.float32, device=scores.device)
    _L.check(_L.load().fo1_topk_desc_f32(scores.data_ptr(), stride, n, k, idx.data_ptr(), val.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "fo1_topk_desc_f32")
    return idx, val


def gather_rows_f32(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    pt, ldt, _, D = _f32(table, "table")
    assert idx.dtype == torch.int32 and idx.is_cuda and idx.is_contiguous()
    out = torch.empty(idx.numel(), D, dtype=torch.float32, device=table.device)
    _L.check(_L.load().fo1_gather_rows_f32(pt, ldt, idx.data_ptr(), out.data_ptr(), out.stride(0), idx.numel(), D, _stream()), "fo1_gather_rows_f32")
    return out


# ---- Swin backbone pieces (swin_ops.hip, attention.hip) ------------------------------------------------------------------------
def attention_window_bias(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, items: torch.Tensor, n_heads: int, head_dim: int, scale: float,
                          bias: torch.Tensor, ws: int, shift: int, nwy: int, nwx: int, flops: float = 0.0) -> torch.Tensor:
    """Swin W-MSA / SW-MSA over consecutive windows of ws*ws tokens (fo1_attention_window_bias_bf16); bias fp32 [heads, ws*ws, ws*ws]."""
    _chk(q, "q"); _chk(k, "k"); _chk(vt, "vt")
    wlen = ws * ws
    if not bias.is_cuda or bias.dtype != torch.float32 or tuple(bias.shape) != (n_heads, wlen, wlen) or not bias.is_contiguous():
        raise TypeError("attention_window_bias: bias must be a contiguous fp32 GPU tensor [heads, ws*ws, ws*ws]")
    pq, ldq, L, _ = _rows(q, "q")
    pk, ldk, _, _ = _rows(k, "k")
    pv, ldv, _, _ = _rows(vt, "vt")
    out = torch.empty(L, n_heads * head_dim, dtype=torch.bfloat16, device=q.device)
    rc = _L.load().fo1_attention_window_bias_bf16(pq, ldq, head_dim, pk, ldk, head_dim, pv, ldv, out.data_ptr(), out.stride(0), head_dim, items.data_ptr(),
                                                  items.shape[0], getattr(items, "q_block", 64), n_heads, head_dim, float(scale), bias.data_ptr(), wlen, ws,
                                                  shift, nwy, nwx, float(flops), _stream())
    _L.check(rc, "fo1_attention_window_bias_bf16")
    return out


def swin_window_partition(x: torch.Tensor, H: int, W: int, ws: int, shift: int, batch: int = 1) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == batch * H * W
    C = x.shape[1]
    nwy, nwx = -(-H // ws), -(-W // ws)
    xw = torch.empty(batch * nwy * nwx * ws * ws, C, dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_swin_window_partition_bf16(x.data_ptr(), xw.data_ptr(), H, W, C, ws, shift, batch, _stream()), "fo1_swin_window_partition_bf16")
    return xw


def swin_window_reverse_add(yw: torch.Tensor, shortcut: torch.Tensor, H: int, W: int, ws: int, shift: int, batch: int = 1) -> torch.Tensor:
    _chk(yw, "yw"); _chk(shortcut, "shortcut")
    assert yw.is_contiguous() and shortcut.is_contiguous() and shortcut.shape[0] == batch * H * W
    y = torch.empty_like(shortcut)
    _L.check(_L.load().fo1_swin_window_reverse_add_bf16(yw.data_ptr(), shortcut.data_ptr(), y.data_ptr(), H, W, shortcut.shape[1], ws, shift, batch, _stream()),
             "fo1_swin_window_reverse_add_bf16")
    return y


def patch_merge(x: torch.Tensor, H: int, W: int, batch: int = 1) -> torch.Tensor:
    _chk(x, "x")
    assert x.is_contiguous() and x.shape[0] == batch * H * W
    C = x.shape[1]
    out = torch.empty(batch * ((H + 1) // 2) * ((W + 1) // 2), 4 * C, dtype=torch.bfloat16, device=x.device)
    _L.check(_L.load().fo1_patch_merge_bf16(x.data_ptr(), out.data_ptr(), H, W, C, batch, _stream()), "fo1_patch_merge_bf16")
    return out


def groupnorm_tokens(x: torch.Tensor, groups: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _chk(x, "x"); _chk(weight, "weight"); _chk(bias, "bias")
    px, ldx, S, C = _rows(x, "x")
    y = torch.empty(S, C, dtype=torch.bfloat16, device=x.device)
    ws = _workspace("groupnorm", x.device, _L.load().fo1_groupnorm_tokens_workspace_bytes(S, groups))
    _L.check(_L.load().fo1_groupnorm_tokens_bf16(px, ldx, S, C, groups, weight.data_ptr(), bias.data_ptr(), float(eps), y.data_ptr(), y.stride(0), ws.data_ptr(),
                                                 ws.numel(), _stream()), "fo1_groupnorm_tokens_bf16")
    return y
