"""Generates tests/golden/msda_ref.npz with the REFERENCE's own pure-torch multi-scale deformable attention
(detect_tools/upn/ops/functions/ms_deform_attn_func.py:41-61, ms_deform_attn_core_pytorch — the function the reference's
ops/test.py checks its CUDA op against), imported in place from /root/reference (this container only).

The module imports the compiled extension `MultiScaleDeformableAttention` at load time (CUDA-only, not built here): an empty stub
module stands in for it — only the pure-torch function is used.  Inputs are not stored: tests/msda_cases.py regenerates them from
seeds (the reference test's own configuration and seed, plus UPN-shaped and ragged cases); outputs of the large cases are the
fp64 evaluation rounded to fp32.

usage: python tests/golden/make_msda_golden.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msda_cases as C  # noqa: E402

REF = "/root/reference/detect_tools/upn/ops/functions/ms_deform_attn_func.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msda_ref.npz")


def reference_function():
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    spec = importlib.util.spec_from_file_location("ref_ms_deform_attn_func", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ms_deform_attn_core_pytorch


def main():
    core = reference_function()
    out = {}
    for tag, (value, shapes, start, loc, w) in C.reference_test_inputs().items():
        out[tag] = core(value, torch.as_tensor(shapes, dtype=torch.long), loc, w).numpy()
    for tag in C.CASES:
        value, shapes, start, loc, w = C.draw(tag)
        out[tag] = core(value.double(), torch.as_tensor(shapes, dtype=torch.long), loc.double(), w.double()).numpy().astype(np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
