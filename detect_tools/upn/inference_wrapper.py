"""Drop-in for the reference's detect_tools/upn/inference_wrapper.py (UPNWrapper :29-237) over the MI355X engine.

Host side (as in the reference, where all of this runs on the CPU): PIL decode + resize to the 800 / 1333 rule
(transforms/transform.py:6-36 + RandomResize([800], max_size=1333), inference_wrapper.py:128-134), box conversion, score sort, score
threshold and NMS.  Device side: rescale / normalise (uint8 upload + table, as the main path's preprocessing) and the whole model.
`torchvision.ops.nms` (inference_wrapper.py:9,217) is restated here in numpy (greedy, boxes with IoU > threshold suppressed, the
published CPU algorithm); torchvision is not installed on either box, so that restatement is unpinned."""
import copy
import os
from typing import Dict, List, Union

import numpy as np
import torch
from PIL import Image

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)         # inference_wrapper.py:132


def nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms semantics on the host: indices of the kept boxes, by decreasing score; a box is dropped when its IoU with an
    already kept box is > iou_threshold.  boxes [n, 4] xyxy."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    keep = []
    suppressed = np.zeros(len(boxes), dtype=bool)
    for i in order:
        if suppressed[i]:
            continue
        keep.append(i)
        xx1, yy1 = np.maximum(x1[i], x1), np.maximum(y1[i], y1)
        xx2, yy2 = np.minimum(x2[i], x2), np.minimum(y2[i], y2)
        inter = np.maximum(xx2 - xx1, 0) * np.maximum(yy2 - yy1, 0)
        iou = inter / (areas[i] + areas - inter)
        suppressed |= iou > iou_threshold
    return np.asarray(keep, dtype=np.int64)


def resize_size(w: int, h: int, size: int = 800, max_size: int = 1333):
    """transforms/transform.py:9-27 (get_size_with_aspect_ratio): -> (oh, ow)."""
    if max_size is not None:
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


class UPNWrapper:
    """A wrapper class for the UPN model (reference docstring: inference_wrapper.py:30-35).

    Args:
        ckpt_path (str): The path to the model checkpoint (`torch.load(...)["model"]` state dict of configs/upn_large.py), or an already
            loaded state dict (tests)."""

    def __init__(self, ckpt_path: Union[str, Dict[str, torch.Tensor]], device: str = "cuda", **engine_kwargs):
        from vlm_fo1_amd.upn import UPNEngine
        if isinstance(ckpt_path, (str, os.PathLike)):
            checkpoint = torch.load(ckpt_path, map_location="cpu")
            state = checkpoint["model"] if "model" in checkpoint else checkpoint
        else:
            state = ckpt_path
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}      # clean_state_dict (utils/detr_utils.py:220-226)
        self.device = torch.device(device)
        self.model = UPNEngine(state, device, **engine_kwargs)
        self.use_graph = True
        self._lut = None

    # ---- reference API ----------------------------------------------------------------------------------------------------------------
    def inference(self, image: List[Union[str, Image.Image]], prompt_type: str = "fine_grained_prompt"):
        """-> {"original_xyxy_boxes": np.ndarray [batch, N, 4] sorted by score, "scores": torch.Tensor [batch, N, 1]} (:42-67)."""
        if not isinstance(image, list):
            image = [image]
        input_images, image_sizes = self.construct_input(image)
        outputs = self._inference(input_images, prompt_type)
        return self.postprocess(outputs, image_sizes)

    def _inference(self, input_images: List[torch.Tensor], prompt_type: str):
        boxes, logits = [], []
        for img in input_images:                          # the engine runs one image per pass (the reference pads a batch into a NestedTensor)
            # one hipGraph per image size from its second sighting on (torch.stack below copies the graph's static outputs)
            out = self.model.forward_graph(img, prompt_type) if self.use_graph else self.model.forward(img, prompt_type)
            boxes.append(out["pred_boxes"].clone())
            logits.append(out["pred_logits"][:, None].clone())
        return dict(pred_boxes=torch.stack(boxes), pred_logits=torch.stack(logits))

    def construct_input(self, image: List[Union[str, Image.Image]]):
        input_images, image_sizes = [], []
        for img in image:
            if isinstance(img, str):
                img = Image.open(img)
            elif not isinstance(img, Image.Image):
                raise ValueError("image must be either a string or a PIL.Image.Image object")
            W, H = img.size
            image_sizes.append([H, W])
            input_images.append(self.transform_image(img))
        return input_images, image_sizes

    def transform_image(self, image_pil: Image.Image) -> torch.Tensor:
        """RandomResize([800], max_size=1333) -> ToTensor -> Normalize (inference_wrapper.py:118-136): [3, h, w] on the device."""
        from vlm_fo1.model.image_processing import normalise_lut
        from vlm_fo1_amd import ops
        img = image_pil.convert("RGB")
        oh, ow = resize_size(*img.size)
        if (ow, oh) != img.size:
            img = img.resize((ow, oh), Image.Resampling.BILINEAR)          # torchvision F.resize on a PIL image: bilinear
        if self._lut is None:
            self._lut = normalise_lut(MEAN, STD).to(self.device)
        u8 = torch.from_numpy(np.array(img, dtype=np.uint8)).to(self.device)
        return ops.normalize_u8(u8, self._lut)

    def postprocess(self, outputs: Dict[str, torch.Tensor], image_pil_sizes: List[List[int]] = None):
        """cxcywh -> xyxy, scale to the original size, sort by score (inference_wrapper.py:138-185; host arithmetic, same op order)."""
        boxes = outputs["pred_boxes"].float().cpu()
        scores = outputs["pred_logits"].float().sigmoid().cpu() if "pred_logits" in outputs else None
        original_xyxy_boxes = []
        for batch_idx, (H, W) in enumerate(image_pil_sizes):
            b = boxes[batch_idx]
            b[:, 0] = b[:, 0] - b[:, 2] / 2
            b[:, 1] = b[:, 1] - b[:, 3] / 2
            b[:, 2] = b[:, 0] + b[:, 2]
            b[:, 3] = b[:, 1] + b[:, 3]
            o = b.clone()
            o[:, 0] *= W
            o[:, 1] *= H
            o[:, 2] *= W
            o[:, 3] *= H
            original_xyxy_boxes.append(o)
        original_xyxy_boxes = torch.stack(original_xyxy_boxes).numpy()
        sorted_boxes, sorted_scores = [], []
        for i in range(len(original_xyxy_boxes)):
            idx = scores[i].squeeze(-1).argsort(descending=True)
            sorted_boxes.append(original_xyxy_boxes[i][idx])
            sorted_scores.append(scores[i][idx])
        return dict(original_xyxy_boxes=np.stack(sorted_boxes), scores=torch.stack(sorted_scores))

    def filter(self, result: Dict, min_score: float, nms_value: float = 0.8):
        """Score threshold + NMS + int32 boxes + 2-decimal scores (inference_wrapper.py:187-237)."""
        filtered_result = {"original_xyxy_boxes": [], "scores": []}
        for boxes, scores in zip(np.array(result["original_xyxy_boxes"]), result["scores"].numpy()):
            keep = scores >= min_score
            boxes = boxes[keep[:, 0]]
            scores = scores[keep[:, 0]][:, 0]
            if len(boxes) == 0:
                return filtered_result
            boxes = boxes.astype(np.float32)
            scores = scores.astype(np.float32)
            keep_indices = nms(boxes, scores, nms_value) if nms_value > 0 else np.arange(len(boxes))
            filtered_boxes = boxes[keep_indices].astype(np.int32)
            filtered_scores = scores[keep_indices]
            order = np.argsort(filtered_scores)[::-1]
            filtered_result["original_xyxy_boxes"].append(filtered_boxes[order].tolist())
            filtered_result["scores"].append([round(float(s), 2) for s in filtered_scores[order]])
        return filtered_result
