"""Host side of the UPN drop-in (detect_tools/upn/inference_wrapper.py): NMS restatement, the 800 / 1333 resize rule, and — when the
reference tree is present (this container) — postprocess() / filter() against the reference's own UPNWrapper methods imported in place
(its torchvision.ops.nms import replaced by this repo's restatement: torchvision is not installed, so nms itself stays unpinned)."""
import os
import sys
import types

import numpy as np
import pytest
import torch


def test_host_nms_known_answers():
    from detect_tools.upn import nms
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], dtype=np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.95], dtype=np.float32)
    assert nms(b, s, 0.5).tolist() == [3, 2]              # IoU(3,0) = 0.952, IoU(3,1) = 0.715 -> both suppressed
    assert nms(b, s, 0.99).tolist() == [3, 0, 1, 2]
    assert nms(b, s, 0.72).tolist() == [3, 1, 2]          # only box 0 (IoU 0.952 > 0.72) goes; box 1 (0.715) stays
    assert nms(np.zeros((0, 4)), np.zeros(0), 0.5).tolist() == []


def test_resize_rule():
    from detect_tools.upn.inference_wrapper import resize_size
    assert resize_size(500, 399) == (800, 1002)            # (w, h) -> (oh, ow): short side to 800
    assert resize_size(640, 480) == (800, 1066)
    assert resize_size(2000, 500) == (333, 1332)           # long side capped at 1333
    assert resize_size(399, 500) == (1002, 800)
    assert resize_size(800, 1000) == (1000, 800)           # already at size


@pytest.mark.skipif(not os.path.isdir("/root/reference/detect_tools/upn"), reason="reference tree not present")
def test_postprocess_and_filter_match_reference_wrapper():
    from detect_tools.upn import inference_wrapper as mine
    # the reference's module, loaded by path with its non-arithmetic imports stubbed
    saved = {k: sys.modules.get(k) for k in ("mmengine", "torchvision", "torchvision.ops", "detect_tools.upn.transforms.transform", "detect_tools.upn.models.module")}
    try:
        sys.modules["mmengine"] = types.SimpleNamespace(Config=None)
        tv = types.ModuleType("torchvision")
        tv.ops = types.ModuleType("torchvision.ops")
        tv.ops.nms = lambda b, s, t: torch.from_numpy(mine.nms(b.numpy(), s.numpy(), t))
        sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, tv.ops
        src = open("/root/reference/detect_tools/upn/inference_wrapper.py").read()
        src = src.replace("import detect_tools.upn.transforms.transform as T", "T = None").replace("from detect_tools.upn import build_architecture", "")
        src = src.replace("from detect_tools.upn.models.module import nested_tensor_from_tensor_list", "")
        ns = {}
        exec(compile(src, "ref_inference_wrapper", "exec"), ns)
        Ref = ns["UPNWrapper"]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    g = torch.Generator().manual_seed(3)
    cxcy = torch.rand(2, 40, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(2, 40, 2, generator=g) * 0.3 + 0.02
    outputs = dict(pred_boxes=torch.cat([cxcy, wh], -1), pred_logits=torch.randn(2, 40, 1, generator=g) * 2)
    sizes = [[480, 640], [399, 500]]
    me = object.__new__(mine.UPNWrapper)
    a = mine.UPNWrapper.postprocess(me, {k: v.clone() for k, v in outputs.items()}, sizes)
    b = Ref.postprocess(None, {k: v.clone() for k, v in outputs.items()}, sizes)
    assert np.array_equal(a["original_xyxy_boxes"], b["original_xyxy_boxes"]) and torch.equal(a["scores"], b["scores"])
    fa = mine.UPNWrapper.filter(me, a, 0.4, 0.5)
    fb = Ref.filter(None, b, 0.4, 0.5)
    assert fa == fb and len(fa["original_xyxy_boxes"]) == 2
    assert mine.UPNWrapper.filter(me, a, 2.0) == Ref.filter(None, b, 2.0)       # nothing above the threshold
