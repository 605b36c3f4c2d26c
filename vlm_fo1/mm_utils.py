"""Host-side prompt / image / box plumbing of the drop-in surface.

Re-implementation — not a copy — of the reference's `vlm_fo1/mm_utils.py`; every public function keeps
the reference's name, signature, return value and observable quirks (cited per function), because
`inference.py`, `scripts/*` and `evaluation/*` call them unmodified.  Equivalence is checked in
tests/test_dropin_surface.py against the reference module imported in place.
"""
from __future__ import annotations

import base64
import io
import random
import re
from typing import Dict, List, Optional, Sequence, Set, Tuple

import torch
from PIL import Image, ImageDraw

from vlm_fo1 import constants as _C
from vlm_fo1.constants import (DEFAULT_REGION_FEATURE_TOKEN, DEFAULT_REGION_INDEX, DEFAULT_REGION_TOKEN,  # noqa: F401
                               IGNORE_INDEX, IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN)

try:  # the stop criterion plugs into HF generate() in the reference; keep the base class when available
    from transformers import StoppingCriteria as _StopBase
except Exception:  # pragma: no cover
    _StopBase = object

# Patched to the Qwen2.5-VL vision markers on the first prepare_inputs() call, exactly like the reference
# mutates its module globals (mm_utils.py:550-553).
DEFAULT_IM_START_TOKEN = _C.DEFAULT_IM_START_TOKEN
DEFAULT_IM_END_TOKEN = _C.DEFAULT_IM_END_TOKEN

_IM_START_ID, _IM_END_ID = 151644, 151645        # <|im_start|>, <|im_end|>  (reference :481-482)
_GROUND_RE = re.compile(r"<ground>(.*?)<\/ground><objects>(.*?)<\/objects>")
_REGION_RE = re.compile(r"<region(\d+)>")


# ------------------------------------------------------------------------------------------------
# tokenisation with sentinels
# ------------------------------------------------------------------------------------------------
def _as_tensor(ids: List[int], return_tensors):
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


_CHUNK_CACHE_MAX = 8192


def _chunk_ids(tokenizer, text: str) -> List[int]:
    """tokenizer(text).input_ids, memoised per tokenizer object.  A region prompt is split into one chunk per
    `<regionfeat>`: the chunks `<region0>`, `<region1>`, ... and the template text recur for every image of a dataset, and the slow
    (`use_fast=False`) tokenizer the reference loads spends ~0.1 ms on each — 100 boxes = 10 ms per image, as much as the GPU pass.
    The ids are a pure function of (tokenizer, text); the cache lives on the tokenizer object and is bounded."""
    try:
        cache = tokenizer.__dict__.setdefault("_fo1_chunk_cache", {})
    except AttributeError:           # objects without a __dict__ (C tokenizers): no cache
        return tokenizer(text).input_ids
    hit = cache.get(text)
    if hit is None:
        if len(cache) >= _CHUNK_CACHE_MAX:
            cache.clear()
        hit = cache[text] = tuple(tokenizer(text).input_ids)
    return list(hit)


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise around `<image>` (or `<image_0>`, `<image_1>`, ... -> always -200): BPE never crosses a
    placeholder; a leading BOS is kept once (reference :29-81)."""
    if "<image_0>" in prompt:
        pieces = re.split(r"<image_[0-9]+>", prompt)
        n_tags = len(re.findall(r"<image_(\d+)>", prompt))
        ids: List[int] = []
        for i, piece in enumerate(pieces):
            ids.extend(_chunk_ids(tokenizer, piece))
            if i < n_tags:
                ids.append(-200)
        return _as_tensor(ids, return_tensors)
    chunks = [_chunk_ids(tokenizer, piece) for piece in prompt.split("<image>")]
    ids = []
    skip = 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        skip = 1
        ids.append(chunks[0][0])
    for i, chunk in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)   # the reference's [sep]*(offset+1) sliced by [offset:] is one id
        ids.extend(chunk[skip:])            # with a BOS-adding tokenizer every chunk starts with BOS: dropped
    return _as_tensor(ids, return_tensors)


def tokenizer_image_region_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX,
                                 region_token_index=DEFAULT_REGION_INDEX, return_tensors=None):
    """Tokenise around `<image>` and `<regionfeat>`: one -200 per image split, one -300 per region split
    (reference :83-135).  When the tokenizer prepends a BOS, it is kept once at the front and stripped from the first text
    chunk of every `<image>` group (the reference's offset quirk, reproduced); chunks after a `<regionfeat>` keep theirs."""
    groups = [[_chunk_ids(tokenizer, part) for part in img_chunk.split("<regionfeat>")] for img_chunk in prompt.split("<image>")]
    ids: List[int] = []
    skip = 0
    if groups and groups[0] and groups[0][0] and groups[0][0][0] == tokenizer.bos_token_id:
        skip = 1
        ids.append(groups[0][0][0])
    last = len(groups) - 1
    for gi, group in enumerate(groups):
        if group:
            ids.extend(group[0][skip:])      # (sic) the BOS offset is applied to the first chunk of EVERY image group
        for chunk in group[1:]:
            ids.append(region_token_index)
            ids.extend(chunk)
        if gi < last:
            ids.append(image_token_index)
    return _as_tensor(ids, return_tensors)


class KeywordsStoppingCriteria(_StopBase):
    """Stop when the generated tail equals a keyword's ids or its decoded text contains the keyword
    (reference :137-181)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids: torch.LongTensor, scores: torch.FloatTensor, **kwargs) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if torch.equal(output_ids[0, -k.shape[0]:], k):
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids: torch.LongTensor, scores: torch.FloatTensor, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))


# ------------------------------------------------------------------------------------------------
# images
# ------------------------------------------------------------------------------------------------
def load_image(image_file):
    """Path / URL / `data:image/...;base64` string -> RGB PIL image of at least 28x28 (reference :183-211;
    a PIL.Image input is accepted here instead of crashing on `.startswith`, Appendix A of SURVEY.md)."""
    if isinstance(image_file, Image.Image):
        image = image_file
    elif image_file.startswith("http"):
        import requests
        image = Image.open(io.BytesIO(requests.get(image_file).content))
    elif image_file.startswith("data:image/"):
        image = Image.open(io.BytesIO(base64.b64decode(image_file.replace("data:image/jpeg;base64,", ""))))
    else:
        image = Image.open(image_file).convert("RGB")
    if image.width < 28 or image.height < 28:
        image = image.resize((max(28, image.width), max(28, image.height)))
    return image


def image_to_base64(img_pil):
    buf = io.BytesIO()
    img_pil.save(buf, format="JPEG")
    return base64.b64encode(buf.getvalue()).decode("utf-8")


def draw_bboxes_and_save(image: Image.Image, fo1_bboxes: dict = {}, detection_bboxes: List[Tuple[int, int, int, int]] = [],
                         output_path: str = "output.jpg", color: str = "red", total_color: str = "green", width: int = 2) -> None:
    """Outline the proposal boxes (total_color) and the labelled FO1 boxes (color + label text), then save
    (reference :230-279)."""
    canvas = ImageDraw.Draw(image)

    def rect(box, outline):
        if len(box) != 4:
            print(f"warning: skipping malformed box {box}")
            return False
        canvas.rectangle([(box[0], box[1]), (box[2], box[3])], outline=outline, width=width)
        return True

    for box in detection_bboxes:
        rect(box, total_color)
    for label, boxes in fo1_bboxes.items():
        for box in boxes:
            if rect(box, color):
                canvas.text((box[0], box[1]), label, fill=color)
    try:
        image.save(output_path)
        print(f"image saved to: {output_path}")
    except IOError as e:
        print(f"error: could not save image to {output_path}: {e}")


# ------------------------------------------------------------------------------------------------
# boxes
# ------------------------------------------------------------------------------------------------
def adjust_bbox(bbox_list, original_h, original_w, resize_h, resize_w):
    """Clamp each xyxy box to the original image, then rescale to the resized image — same operation
    order as the reference (`v * resize / original`, :281-312) so the floats are identical."""
    out = []
    for x1, y1, x2, y2 in bbox_list:
        x1, x2 = (max(0, min(original_w, v)) for v in (x1, x2))
        y1, y2 = (max(0, min(original_h, v)) for v in (y1, y2))
        out.append([x1 * resize_w / original_w, y1 * resize_h / original_h, x2 * resize_w / original_w, y2 * resize_h / original_h])
    return out


def extract_predictions_to_indexes(prediction: str) -> Dict[str, Set[int]]:
    """`<ground>label</ground><objects><region3><region7></objects>` -> {label: {3, 7}}; repeated labels are
    unioned (reference :346-369)."""
    found: Dict[str, Set[int]] = {}
    for label, body in _GROUND_RE.findall(prediction):
        label = label.strip()
        idx = {int(n) for n in _REGION_RE.findall(body)}
        found[label] = found[label] | idx if label in found else idx
    return found


def extract_predictions_to_bboxes(prediction: str, bbox_list):
    """Same parse, mapped to the boxes themselves in set-iteration order (reference :314-344)."""
    return {label: [bbox_list[i] for i in idx] for label, idx in extract_predictions_to_indexes(prediction).items()}


def resize_shortest_edge_images_and_bboxes(image_list: List[Image.Image], bbox_lists: List, candidate_sizes: List[int] = [],
                                           max_size: int = 2048):
    """Optionally bring the short edge to a random candidate size, cap the long edge at `max_size`, keep at
    least 28 px per side (bicubic), and scale the boxes by new/old per axis (reference :371-462).  A single
    [N,4] box list is accepted and returned un-nested."""
    single = False
    probe = torch.tensor(bbox_lists)
    if probe.dim() == 2 and probe.shape[1] == 4:
        bbox_lists = [bbox_lists]
        single = True
    if not image_list or not bbox_lists:
        raise ValueError("Input lists cannot be empty.")
    if len(image_list) != len(bbox_lists):
        raise ValueError("The lengths of the image list and the bounding box list must be the same.")
    target = random.choice(candidate_sizes) if len(candidate_sizes) > 0 else None
    images, boxes_out = [], []
    for img, boxes in zip(image_list, bbox_lists):
        w0, h0 = img.size
        scale = target / min(w0, h0) if target else 1.0
        h1, w1 = int(h0 * scale), int(w0 * scale)
        if max(h1, w1) > max_size:
            shrink = max_size / max(h1, w1)
            h1, w1 = int(h1 * shrink), int(w1 * shrink)
        w1, h1 = max(28, w1), max(28, h1)
        images.append(img if (w1, h1) == (w0, h0) else img.resize((w1, h1), Image.Resampling.BICUBIC))
        rx, ry = w1 / w0, h1 / h0
        boxes_out.append([[x1 * rx, y1 * ry, x2 * rx, y2 * ry] for x1, y1, x2, y2 in boxes])
    return (images, boxes_out[0]) if single else (images, boxes_out)


# ------------------------------------------------------------------------------------------------
# prompt assembly
# ------------------------------------------------------------------------------------------------
def make_message_context(tokenizer, message, chat_format="chatml"):
    """One chat message -> (prompt text, token ids with sentinels, image urls, bbox_list)  (reference :464-528)."""
    image_urls: List[str] = []
    if chat_format != "chatml":
        return None
    role, content = message["role"], message["content"]
    bbox_list = message.get("bbox_list", None)
    inp, tokens = None, None
    newline = tokenizer.encode("\n")

    def plain(text):
        body = tokenizer.encode(role, allowed_special=set()) + newline + tokenizer.encode(text, allowed_special=set())
        return f"<|im_start|>{role}\n{text}<|im_end|>\n", [_IM_START_ID] + body + [_IM_END_ID]

    if role == "system":
        inp, tokens = plain(content)
    if role == "user":
        if isinstance(content, str):
            inp, tokens = plain(content)
        if isinstance(content, list):
            has_boxes = bool(bbox_list) and len(bbox_list) > 0
            parts = [f"<|im_start|>{role}\n"]
            for part in content:
                if part["type"] == "text":
                    parts.append(f"{part['text']}")
                if part["type"] == "image_url":
                    parts.append(DEFAULT_IM_START_TOKEN + "<image>" + DEFAULT_IM_END_TOKEN + "\n")
                    if has_boxes:
                        parts.extend(DEFAULT_REGION_TOKEN.replace("<i>", str(i)) + DEFAULT_REGION_FEATURE_TOKEN
                                     for i in range(len(bbox_list)))
                        parts.append("\n")
                    image_urls.append(part["image_url"]["url"])
            parts.append("<|im_end|>\n")
            inp = "".join(parts)
            tokens = tokenizer_image_region_token(inp, tokenizer) if has_boxes else \
                tokenizer_image_token(inp, tokenizer, image_token_index=IMAGE_TOKEN_INDEX)
    return inp, tokens, image_urls, bbox_list


def prepare_inputs(model_name, model, image_processors, tokenizer, messages, device="cuda", max_tokens=512, top_p=1.0,
                   temperature=0.0, do_sample=False):
    """messages -> the kwargs dict for `model.generate(**kwargs)` with exactly the reference's keys
    (:640-654): inputs, images, images_aux, image_grid_thws, bbox_list, do_sample, temperature,
    max_new_tokens, streamer, top_p, use_cache, stopping_criteria, pad_token_id."""
    global DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN
    lowered = model_name.lower()
    if "qwen2.5-vl" in lowered or "qwen2_5_vl" in lowered:
        DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<|vision_start|>", "<|vision_end|>"
    primary_ip, aux_ip = image_processors

    prompt, ids, image_urls, bbox_list = "", [], [], None
    for message in messages:
        text, toks, image_urls, bbox_list = make_message_context(tokenizer, message)
        prompt += text
        ids.extend(toks)
    if "system" not in prompt:   # (sic) a substring test on the whole prompt, reference :567
        sys_text = "system\nYou are a helpful assistant."
        prompt = "<|im_start|>" + sys_text + "<|im_end|>" + "\n" + prompt
        ids = [_IM_START_ID] + tokenizer(sys_text).input_ids + [_IM_END_ID] + tokenizer("\n").input_ids + ids
    if not prompt.endswith("<|im_start|>assistant"):
        prompt += "<|im_start|>" + "assistant" + "\n"
        ids.extend([_IM_START_ID] + tokenizer("assistant\n").input_ids)

    aux_tensors = None
    images: List[Image.Image] = []
    if image_urls:
        images = [load_image(u) for u in image_urls]
        images, bbox_list = resize_shortest_edge_images_and_bboxes(images, bbox_list, max_size=2048)
        if getattr(model.config, "mm_use_region_index_token", False):
            sizes = [im.size for im in images]
            aux_tensors = [aux_ip.preprocess(im, return_tensors="pt")["pixel_values"][0].to(device) for im in images]
            if bbox_list and len(bbox_list) > 0:
                bbox_list = bbox_list[:100]                                 # reference :600
                rh, rw = aux_tensors[0].shape[-2:]
                ow, oh = sizes[0]
                bbox_list = [torch.tensor(adjust_bbox(bbox_list, oh, ow, rh, rw))]
            else:
                bbox_list = None

    pix, grids = [], []
    for im in images:
        data = primary_ip.preprocess(im, videos=None, return_tensors="pt")
        pix.append(data["pixel_values"].to(device))
        grids.append(data["image_grid_thw"])

    if "qwen" in lowered:
        input_ids = torch.tensor([ids]).to(device)
        input_ids._fo1_ids = list(ids)      # host copy for the engine's splice planner (FO1ForCausalLM._request): no device -> host read per request
        keywords = ["<|im_end|>"]
    stopping = KeywordsStoppingCriteria(keywords, tokenizer, input_ids)
    try:
        from transformers import TextStreamer
        streamer = TextStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True)
    except Exception:  # pragma: no cover
        streamer = None
    print("question:================\n", prompt, "\n=================")
    return dict(inputs=input_ids, images=pix, images_aux=aux_tensors, image_grid_thws=grids, bbox_list=bbox_list,
                do_sample=temperature != 0.0, temperature=temperature, max_new_tokens=max_tokens, streamer=streamer,
                top_p=top_p, use_cache=True, stopping_criteria=[stopping], pad_token_id=tokenizer.pad_token_id)
