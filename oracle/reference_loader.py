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


def vendored_llm(**cfg_kwargs):
    """The reference's OWN vendored `Qwen2_5_VLModel` (the LLM half of modeling_qwen2_5_vl.py:1097-1242: decoder layers :1014-1095,
    mRoPE :603-685, attention :738-1004) built on the CPU with sdpa attention.  Two shims for the installed transformers 5, neither of
    which touches the arithmetic under test:
      * `ROPE_INIT_FUNCTIONS["default"]` — transformers 5 dropped the key the vendored rotary module looks up (:578); the shim is the
        published default initialiser, inv_freq = theta^(-2i/d), attention scaling 1;
      * `config.pad_token_id = None` — read by `nn.Embedding(padding_idx=...)` at :1105, absent from the new config class.
    cfg_kwargs: Qwen2_5_VLConfig text fields (vocab_size, hidden_size, ..., rope_scaling={"type": "mrope", "mrope_section": [...]})."""
    import torch
    ref = vendored_qwen()

    def _default_rope(config, device=None, **kw):
        dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        inv = 1.0 / (config.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
        return inv, 1.0

    ref.ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)
    cfg = ref.Qwen2_5_VLConfig(**cfg_kwargs)
    cfg.pad_token_id = None
    cfg._attn_implementation = "sdpa"
    return ref.Qwen2_5_VLModel(cfg).eval()


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


def vendored_vit_encoder():
    """qwen2_5_vl_encoder.py (custom_forward + VisionFeaturesGather) executed in place, with stubs for
    the two imports that need torchvision (ToPILImage, the HF image processor)."""
    if "vit_enc" not in _cache:
        qwen = vendored_qwen()
        saved = {k: sys.modules.get(k) for k in (
            "torchvision", "torchvision.transforms", "vlm_fo1", "vlm_fo1.model", "vlm_fo1.model.multimodal_encoder",
            "vlm_fo1.model.multimodal_encoder.qwen2_5_vl", "vlm_fo1.model.multimodal_encoder.qwen2_5_vl.modeling_qwen2_5_vl",
            "transformers.models.qwen2_vl.image_processing_qwen2_vl")}
        try:
            tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
            tvt.ToPILImage = object
            tv.transforms = tvt
            sys.modules["torchvision"] = tv
            sys.modules["torchvision.transforms"] = tvt
            ip = types.ModuleType("transformers.models.qwen2_vl.image_processing_qwen2_vl")
            ip.Qwen2VLImageProcessor = object
            sys.modules["transformers.models.qwen2_vl.image_processing_qwen2_vl"] = ip
            for pkg in ("vlm_fo1", "vlm_fo1.model", "vlm_fo1.model.multimodal_encoder", "vlm_fo1.model.multimodal_encoder.qwen2_5_vl"):
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
            sys.modules["vlm_fo1.model.multimodal_encoder.qwen2_5_vl.modeling_qwen2_5_vl"] = qwen
            _cache["vit_enc"] = _load("ref_qwen_vit_encoder", os.path.join(
                REFERENCE_ROOT, "vlm_fo1", "model", "multimodal_encoder", "qwen2_5_vl_encoder.py"))
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
    return _cache["vit_enc"]
