"""Synthetic stand-ins for what neither box has (no checkpoint, tokenizer or dataset offline — SURVEY §8c/§8d): a deterministic
tokenizer, the checkpoint's config.json at the true Qwen2.5-VL-3B + DaViT-L shapes, and a COCO-eval-shaped dataset on disk (JPEG files +
the jsonl / instances json evaluation/eval_coco.py reads).  Used by bench.py's `driver_level` block and by the tests."""
from __future__ import annotations

import json
import os
import types

COCO_NAMES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant", "stop sign",
              "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack",
              "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat"]


class ToyTokenizer:
    """Deterministic stand-in (no checkpoint offline): words -> hashed ids, optional BOS; decode -> the ids as text."""
    pad_token_id = 0

    def __init__(self, bos=None, vocab: int = 5000, offset: int = 10):
        self.bos_token_id = bos
        self.vocab, self.offset = vocab, offset

    def _enc(self, text):
        ids = [(sum(ord(c) * (i + 7) for i, c in enumerate(w)) % self.vocab) + self.offset for w in text.replace("\n", " \n ").split(" ") if w != ""]
        return ([self.bos_token_id] if self.bos_token_id is not None else []) + ids

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=self._enc(text))

    def encode(self, text, allowed_special=None):
        return self._enc(text)

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(i)) for i in row) for row in ids]

    def decode(self, ids, **kw):
        return " ".join(str(int(i)) for i in ids)


def full_config_dict() -> dict:
    """config.json of the released VLM-FO1_Qwen2.5-VL-3B as far as the tree lets one know it (Qwen2.5-VL-3B text / vision
    configs, DaViT-L aux tower, SimpleFPN + concat region features of width 5888, mlp2x_gelu connectors)."""
    return {
        "model_type": "omchat_qwen2_5_vl", "hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16,
        "num_key_value_heads": 2, "intermediate_size": 11008, "vocab_size": 151936, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
        "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "tie_word_embeddings": True, "eos_token_id": 151645,
        "vision_config": {"depth": 32, "hidden_size": 1280, "num_heads": 16, "intermediate_size": 3420, "out_hidden_size": 2048,
                          "patch_size": 14, "spatial_merge_size": 2, "temporal_patch_size": 2, "window_size": 112,
                          "fullatt_block_indexes": [7, 15, 23, 31]},
        "mm_vision_tower": "qwen2.5-vl", "mm_vision_tower_aux": "davit-large", "mm_projector_type": "mlp2x_gelu",
        "mm_projector_aux_type": "mlp2x_gelu", "mm_use_vision_tower_region_feature": True, "mm_use_simpleFPN_for_vt": True,
        "mm_region_hidden_size": 5888, "mm_use_region_index_token": True, "aux_image_size": 768, "aux_image_aspect_ratio": "dynamic",
    }


def write_coco_like_dataset(root: str, n_items: int, boxes_per_item: int = 100, size=(640, 480), seed: int = 1234, quality: int = 90):
    """n_items JPEG files of `size` (smooth noise: decodes at a photo's cost, not a flat colour's) + <root>/eval.jsonl in the format
    of the reference's processed COCO file (evaluation/eval_coco.py:20-35: id, image, conversations[0].value, bbox_list, score_list) +
    <root>/instances.json (categories).  Boxes: the CountBench / Pixmo UPN fixture boxes rescaled to the image, as bench.py's main
    workload.  -> (eval jsonl path, instances json path, image folder)."""
    import numpy as np
    from PIL import Image
    from vlm_fo1.task_templates import OD_template
    from .hfre_cases import box_fixtures
    W, H = size
    img_dir = os.path.join(root, "images")
    os.makedirs(img_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    fx = box_fixtures()
    pools = sorted(fx["countbench"] + fx["pixmo"], key=lambda x: -len(x["bboxes"]))
    question = OD_template.format(", ".join(COCO_NAMES))
    lines = []
    for i in range(n_items):
        small = rng.integers(0, 256, (H // 8, W // 8, 3), dtype=np.uint8)
        im = Image.fromarray(small, "RGB").resize((W, H), Image.BICUBIC)
        name = f"img{i:06d}.jpg"
        im.save(os.path.join(img_dir, name), quality=quality)
        it = pools[i % max(1, sum(1 for p in pools if len(p["bboxes"]) >= boxes_per_item))]
        ex, ey = it["extent"]
        boxes = [[b[0] * W / ex, b[1] * H / ey, b[2] * W / ex, b[3] * H / ey] for b in it["bboxes"][:boxes_per_item]]
        lines.append(json.dumps({"id": i, "image": name, "conversations": [{"from": "human", "value": question}], "bbox_list": boxes,
                                 "score_list": [round(0.9 - 0.005 * k, 4) for k in range(len(boxes))]}))
    eval_path = os.path.join(root, "eval.jsonl")
    with open(eval_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    inst = os.path.join(root, "instances.json")
    json.dump({"categories": [{"name": n, "id": k + 1} for k, n in enumerate(COCO_NAMES)]}, open(inst, "w"))
    return eval_path, inst, img_dir
