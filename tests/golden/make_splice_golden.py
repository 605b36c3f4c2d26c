"""Generates tests/golden/splice_ref.npz: what the REFERENCE's own `prepare_inputs_labels_for_qwen2_5_vl_multimodal`
(vlm_fo1/model/language_model/omchat_qwen2_5_vl.py:135-463: sentinel splice, right padding, get_rope_index) returns on the seeded
prompts of tests/splice_cases.py — run in place, on the CPU, as an unbound method on a stand-in `self` whose encode_images /
encode_regions hand back the seeded feature tensors (the function itself is pure indexing).

The reference package is imported under its own name `vlm_fo1` (this repo's drop-in package of the same name is kept off sys.path),
with import-time stubs for packages that are not installed (torchvision, timm, the torchvision-based HF image processor): none of them
is touched by the function under test.

    python tests/golden/make_splice_golden.py
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT and os.path.abspath(p or ".") != HERE]
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.join(ROOT, "tests"))          # splice_cases.py only

import numpy as np          # noqa: E402
import torch                # noqa: E402
import transformers         # noqa: E402,F401  (before the stubs: its availability probes must see the really absent torchvision)


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _nope(*a, **k):
    raise RuntimeError("stub: not on the path under test")


tv = stub("torchvision"); tv.__path__ = []
tv.ops = stub("torchvision.ops", roi_align=_nope)
tv.transforms = stub("torchvision.transforms", ToPILImage=object, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
stub("torchvision.transforms.functional")
stub("transformers.models.qwen2_vl.image_processing_qwen2_vl", Qwen2VLImageProcessor=object)
timm = stub("timm"); timm.__path__ = []
tm = stub("timm.models"); tm.__path__ = []
tm.layers = stub("timm.models.layers", DropPath=torch.nn.Identity, trunc_normal_=_nope)
timm.layers = stub("timm.layers", LayerNorm=torch.nn.LayerNorm, LayerNorm2d=torch.nn.LayerNorm, DropPath=torch.nn.Identity, trunc_normal_=_nope)
stub("timm.models.regnet", RegStage=object)
stub("timm.models.resnet", Bottleneck=object)

import vlm_fo1.model.language_model.omchat_qwen2_5_vl as O      # noqa: E402  (the reference's)
import splice_cases as C                                        # noqa: E402

assert O.__file__.startswith("/root/reference/"), O.__file__


def run(batch):
    tower = object.__new__(O.Qwen2_5_VlVisionTower)              # isinstance() checks at :160,192,433 — no __init__, no weights
    torch.nn.Module.__init__(tower)
    tower.is_loaded = False
    tower.cfg_only = types.SimpleNamespace(patch_size=14)
    emb = torch.nn.Embedding(C.VOCAB, C.D)
    emb.weight.data.copy_(C.embed_table())
    model = types.SimpleNamespace(embed_tokens=emb, get_vision_tower=lambda: tower)
    cfg = types.SimpleNamespace(image_token_id=C.IMAGE_TOKEN_ID, video_token_id=151656, vision_start_token_id=C.VISION_START, bos_token_id=C.BOS,
                                tokenizer_padding_side="right", tokenizer_model_max_length=None,
                                vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=2))
    grids = [torch.tensor([[1, p["grid_merged"][0] * 2, p["grid_merged"][1] * 2]]) for p in batch]
    fake = types.SimpleNamespace(config=cfg, device=torch.device("cpu"), get_vision_tower=lambda: tower, get_video_tower=lambda: None,
                                 get_vision_tower_aux=lambda: object(), get_model=lambda: model,
                                 encode_images=lambda images, thw: ([p["image_tokens"] for p in batch], grids, [None] * len(batch)),
                                 encode_regions=lambda *a, **k: [p["region_tokens"][:max(p["n_regions"], 1)] for p in batch])
    fake.get_rope_index = types.MethodType(O.Qwen2_5_VLForConditionalGeneration.get_rope_index, fake)
    L = max(len(p["ids"]) for p in batch)
    ids = torch.full((len(batch), L), 0, dtype=torch.long)
    mask = torch.zeros(len(batch), L, dtype=torch.long)
    for i, p in enumerate(batch):
        ids[i, :len(p["ids"])] = torch.tensor(p["ids"])
        mask[i, :len(p["ids"])] = 1
    images = [torch.zeros(4, 1176) for _ in batch]              # 2-D "pixel_values" placeholders (ndim check at :157)
    bbox = [torch.zeros(max(p["n_regions"], 1), 4) for p in batch]
    with torch.no_grad():
        out = O.OmChatQwen25VLForCausalLM.prepare_inputs_labels_for_qwen2_5_vl_multimodal(
            fake, ids, None, mask, None, None, images, images_aux=[torch.zeros(3, 8, 8) for _ in batch], bbox_list=bbox, image_grid_thws=grids)
    _, position_ids, attention_mask, _, embeds, _, rope_deltas, cache_position = out
    return dict(embeds=embeds.numpy(), position_ids=position_ids.numpy(), attention_mask=attention_mask.numpy(),
                rope_deltas=rope_deltas.numpy(), cache_position=cache_position.numpy())


if __name__ == "__main__":
    out = {}
    for name, batch in C.cases().items():
        r = run(batch)
        for k, v in r.items():
            out[f"{name}.{k}"] = v
        print(name, r["embeds"].shape, r["position_ids"].shape, r["rope_deltas"].ravel().tolist())
    np.savez_compressed(os.path.join(HERE, "splice_ref.npz"), **out)
    print("wrote tests/golden/splice_ref.npz")
