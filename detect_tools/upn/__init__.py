"""detect_tools.upn — the UPN proposal detector on MI355X (drop-in for the reference's detect_tools/upn package surface that its
drivers touch: `UPNWrapper`, inference.py:3 / scripts/inference_with_upn.py:4-9,22-40).

`UPNWrapper(ckpt_path)` loads the reference's checkpoint format (`torch.load(ckpt)["model"]`, inference_wrapper.py:16-26) into the
hand-written gfx950 engine (vlm_fo1_amd/upn.py: Swin-L backbone, deformable encoder / decoder over the MSDA kernel), and keeps the
reference's `inference(image, prompt_type)` / `filter(result, min_score, nms_value)` contract (inference_wrapper.py:42-237)."""
from .inference_wrapper import UPNWrapper, nms  # noqa: F401

__all__ = ["UPNWrapper", "nms"]
