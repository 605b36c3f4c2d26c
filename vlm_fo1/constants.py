"""Sentinel ids and special-token strings of the VLM-FO1 prompt format.

Interface constants: values must equal the reference's vlm_fo1/constants.py:1-29 (checked by
tests/test_dropin_surface.py when the reference tree is present)."""

_INT = {
    "IGNORE_INDEX": -100,               # label padding
    "IMAGE_TOKEN_INDEX": -200,          # sentinel id standing for all image tokens of one image
    "DEFAULT_REGION_INDEX": -300,       # sentinel id standing for one region-feature token
    "QWEN2_5_VL_IMAGE_TOKEN_INDEX": 151655,
}
_STR = {
    "LOGDIR": ".",
    "DEFAULT_IMAGE_TOKEN": "<image>",
    "DEFAULT_IMAGE_PATCH_TOKEN": "<im_patch>",
    "DEFAULT_IM_START_TOKEN": "<im_start>",
    "DEFAULT_IM_END_TOKEN": "<im_end>",
    "QWEN2_5_VL_IMAGE_TOKEN": "<|image_pad|>",
    "DEFAULT_REGION_TOKEN": "<region<i>>",           # "<i>" is replaced by the box index
    "DEFAULT_REGION_FEATURE_TOKEN": "<regionfeat>",
    "DEFAULT_GROUNDING_START": "<ground>",
    "DEFAULT_GROUNDING_END": "</ground>",
    "DEFAULT_GROUNDING_OBJECTS_START": "<objects>",
    "DEFAULT_GROUNDING_OBJECTS_END": "</objects>",
    "DEFAULT_THINK_START": "<think>",
    "DEFAULT_THINK_END": "</think>",
}
globals().update(_INT)
globals().update(_STR)
__all__ = sorted(list(_INT) + list(_STR))
