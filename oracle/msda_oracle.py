"""ORACLE — test infrastructure only (never imported by the product path).

Multi-scale deformable attention forward on the CPU: ctypes front end of oracle/msda_ref.c, the C restatement of the reference's
CUDA kernel (detect_tools/upn/ops/src/cuda/ms_deform_im2col_cuda.cuh:32-84,237-299).  Pinned against the reference's own
ms_deform_attn_core_pytorch (detect_tools/upn/ops/functions/ms_deform_attn_func.py:41-61) through tests/golden/msda_ref.npz
(tests/golden/make_msda_golden.py)."""
import ctypes

import numpy as np
import torch

from .build import build

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name, ct in (("msda_forward_f32", ctypes.c_float), ("msda_forward_f64", ctypes.c_double)):
            fn = getattr(_lib, name)
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    return _lib


def ms_deform_attn_forward(value: torch.Tensor, spatial_shapes, level_start_index, sampling_locations: torch.Tensor,
                           attention_weights: torch.Tensor) -> torch.Tensor:
    """value [N, S, M, D], sampling_locations [N, Lq, M, L, P, 2], attention_weights [N, Lq, M, L, P] (float32 or float64, CPU)
    -> [N, Lq, M*D], the layout MSDeformAttnFunction.forward returns (ms_deform_attn_cuda.cu:77)."""
    dt = value.dtype
    assert dt in (torch.float32, torch.float64) and sampling_locations.dtype == dt and attention_weights.dtype == dt
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    npdt = np.float32 if dt == torch.float32 else np.float64
    v = np.ascontiguousarray(value.numpy(), dtype=npdt)
    loc = np.ascontiguousarray(sampling_locations.numpy(), dtype=npdt)
    w = np.ascontiguousarray(attention_weights.numpy(), dtype=npdt)
    sh = np.ascontiguousarray(np.asarray(spatial_shapes, dtype=np.int64).reshape(L, 2))
    ls = np.ascontiguousarray(np.asarray(level_start_index, dtype=np.int64).reshape(L))
    assert int((sh[:, 0] * sh[:, 1]).sum()) == S
    out = np.zeros((N, Lq, M * D), dtype=npdt)
    fn = _load().msda_forward_f32 if dt == torch.float32 else _load().msda_forward_f64
    fn(v.ctypes.data, sh.ctypes.data, ls.ctypes.data, loc.ctypes.data, w.ctypes.data, N, S, M, D, L, Lq, P, out.ctypes.data)
    return torch.from_numpy(out)
