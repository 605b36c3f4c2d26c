"""ORACLE — test infrastructure only.  Imports the reference's own modules IN PLACE from
/root/reference (this container only; never copied) so the restatements can be pinned."""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_QWEN_DIR = os.path.join(REFERENCE_ROOT, "vlm_fo1", "model", "multimodal_encoder", "qwen2_5_vl")
_DAVIT_DIR = os.path.join(REFERENCE_ROOT, "vlm_fo1", "model", "multimodal_encoder", "davit")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vlm_fo1"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def vendored_qwen():
    """The vendored modeling_qwen2_5_vl.py as a standalone package `ref_qwen2_5_vl` (its ViT half and
    the pure functions — get_rope_index, rotary helpers — work under the installed transformers)."""
    if "qwen" not in _cache:
        import transformers  # noqa: F401
        pkg = types.ModuleType("ref_qwen2_5_vl")
        pkg.__path__ = [_QWEN_DIR]
        sys.modules["ref_qwen2_5_vl"] = pkg
        _load("ref_qwen2_5_vl.configuration_qwen2_5_vl", os.path.join(_QWEN_DIR, "configuration_qwen2_5_vl.py"))
        _cache["qwen"] = _load("ref_qwen2_5_vl.modeling_qwen2_5_vl", os.path.join(_QWEN_DIR, "modeling_qwen2_5_vl.py"))
    return _cache["qwen"]


def vendored_davit():
    """The reference DaViT (modeling_davit.py) with a 2-symbol shim for timm (init/regulariser only)."""
    if "davit" not in _cache:
        import transformers  # noqa: F401
        import torch
        if "timm" not in sys.modules:
            timm = types.ModuleType("timm")
            tm = types.ModuleType("timm.models")
            tl = types.ModuleType("timm.models.layers")

            class DropPath(torch.nn.Module):  # identity at inference
                def __init__(self, p=0.0):
                    super().__init__()

                def forward(self, x):
                    return x

            tl.DropPath = DropPath
            tl.trunc_normal_ = torch.nn.init.trunc_normal_
            timm.models = tm
            tm.layers = tl
            sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
        _cache["davit"] = _load("ref_davit_modeling", os.path.join(_DAVIT_DIR, "modeling_davit.py"))
        _cache["davit_cfg"] = _load("ref_davit_configs", os.path.join(_DAVIT_DIR, "configs.py"))
    return _cache["davit"], _cache["davit_cfg"]
