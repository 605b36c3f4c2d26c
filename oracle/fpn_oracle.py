"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement (plain torch fp32 on bf16-valued weights) of the ViTDet SimpleFPN the HFRE applies
to the last captured ViT map when `mm_use_simpleFPN_for_vt` is set:
  vlm_fo1/model/multimodal_visual_prompt_encoder/simple_fpn.py  Conv2d :28-56, LayerNorm :58-78,
  SimpleFP :100-216  (instantiated at hybrid_finegrained_region_encoder.py:175 with
  out_channels=512, norm="LN", dim=1280, stride=14 -> stages simfp_1..simfp_4, no conv bias).
Pinned in tests/test_oracle_fpn.py against the reference module imported in place.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

STAGES = ("simfp_1", "simfp_2", "simfp_3", "simfp_4")  # scale 4, 2, 1, 0.5 -> strides 3.5, 7, 14, 28


def _cln(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w.float()[:, None, None] * x + b.float()[:, None, None]


def fpn_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """x [1, dim, H, W] -> 4 maps [1, 512, 4H.., 2H.., H.., H/2..] (fp32)."""
    x = x.float()
    outs = []
    for si, name in enumerate(STAGES):
        p = name + "."
        h = x
        if si == 0:
            h = F.conv_transpose2d(h, sd[p + "0.weight"].float(), sd[p + "0.bias"].float(), stride=2)
            h = F.gelu(_cln(h, sd[p + "1.weight"], sd[p + "1.bias"]))
            h = F.conv_transpose2d(h, sd[p + "3.weight"].float(), sd[p + "3.bias"].float(), stride=2)
            a, b = "4.", "5."
        elif si == 1:
            h = F.conv_transpose2d(h, sd[p + "0.weight"].float(), sd[p + "0.bias"].float(), stride=2)
            a, b = "1.", "2."
        elif si == 2:
            a, b = "0.", "1."
        else:
            h = F.max_pool2d(h, 2, 2)
            a, b = "1.", "2."
        h = _cln(F.conv2d(h, sd[p + a + "weight"].float()), sd[p + a + "norm.weight"], sd[p + a + "norm.bias"])
        h = _cln(F.conv2d(h, sd[p + b + "weight"].float(), padding=1), sd[p + b + "norm.weight"], sd[p + b + "norm.bias"])
        outs.append(h)
    return outs


def random_fpn_state(dim=1280, out=512, seed=0, std=0.02):
    g = torch.Generator().manual_seed(seed)

    def w(*s, sc=std):
        return (torch.randn(*s, generator=g) * sc).bfloat16()

    def ln(sd, p, c):
        sd[p + "weight"] = (1 + 0.1 * torch.randn(c, generator=g)).bfloat16()
        sd[p + "bias"] = (0.05 * torch.randn(c, generator=g)).bfloat16()

    sd = {}
    sd["simfp_1.0.weight"], sd["simfp_1.0.bias"] = w(dim, dim // 2, 2, 2), w(dim // 2, sc=0.05)
    ln(sd, "simfp_1.1.", dim // 2)
    sd["simfp_1.3.weight"], sd["simfp_1.3.bias"] = w(dim // 2, dim // 4, 2, 2), w(dim // 4, sc=0.05)
    sd["simfp_2.0.weight"], sd["simfp_2.0.bias"] = w(dim, dim // 2, 2, 2), w(dim // 2, sc=0.05)
    for name, a, b, cin in (("simfp_1", "4.", "5.", dim // 4), ("simfp_2", "1.", "2.", dim // 2), ("simfp_3", "0.", "1.", dim),
                            ("simfp_4", "1.", "2.", dim)):
        sd[f"{name}.{a}weight"] = w(out, cin, 1, 1)
        ln(sd, f"{name}.{a}norm.", out)
        sd[f"{name}.{b}weight"] = w(out, out, 3, 3)
        ln(sd, f"{name}.{b}norm.", out)
    return sd
