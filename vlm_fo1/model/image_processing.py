"""Host-side image preprocessing for the two towers, dependency-free (PIL + numpy):
no torchvision, no HF image-processor classes (the default one needs torchvision under transformers 5).

* Qwen2VLPatchProcessor — what the reference obtains from `Qwen2VLImageProcessor.from_pretrained(model_path,
  min_pixels=56*56, max_pixels=2048*2048)` (qwen2_5_vl_encoder.py:179,210): smart-resize to multiples of 28
  within [min_pixels, max_pixels], bicubic, /255, CLIP mean/std, then patches in 2x2 merge-block order, each
  patch vector (C=3, T=2, 14, 14) with the single frame duplicated along T.  -> pixel_values [S, 1176],
  image_grid_thw [1, 3].
* CLIPStyleAuxProcessor — the reference's `CLIPImageProcessor(**img_cfg)` (davit/configs.py:139-152,
  image_processing_clip.py:222-367): RGB, optional squash-resize to 768x768 (bicubic) — skipped in 'dynamic'
  mode — /255, ImageNet mean/std, channels first.
Equivalence with the installed HF PIL processor / the reference CLIP processor: tests/test_dropin_surface.py."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
from PIL import Image

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 2048 * 2048):
    """Both sides to multiples of `factor`, area clamped to [min_pixels, max_pixels], aspect kept."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    hb, wb = round(height / factor) * factor, round(width / factor) * factor
    if hb * wb > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        hb = max(factor, math.floor(height / beta / factor) * factor)
        wb = max(factor, math.floor(width / beta / factor) * factor)
    elif hb * wb < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        hb = math.ceil(height * beta / factor) * factor
        wb = math.ceil(width * beta / factor) * factor
    return hb, wb


def normalise_lut(mean, std) -> torch.Tensor:
    """bf16 [3, 256]: every possible uint8 sample through `_normalise`'s arithmetic, then the tower's bf16 cast — the whole
    per-sample computation as a table for the device path (fo1_patchify_u8_bf16 / fo1_normalize_u8_bf16)."""
    v = np.arange(256, dtype=np.uint8)
    a = (v.astype(np.float64) * (1 / 255)).astype(np.float32)
    a = (a[None, :] - np.array(mean, dtype=np.float32)[:, None]) / np.array(std, dtype=np.float32)[:, None]
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)


def _normalise(img: Image.Image, mean, std) -> np.ndarray:
    """uint8 HWC -> float32 CHW, (x/255 - mean)/std with HF's rounding points (rescale in float64 -> float32,
    normalise in float32)."""
    a = np.asarray(img, dtype=np.uint8)
    a = (a.astype(np.float64) * (1 / 255)).astype(np.float32)
    a = (a - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32)
    return np.ascontiguousarray(a.transpose(2, 0, 1))


class Qwen2VLPatchProcessor:
    def __init__(self, min_pixels: int = 56 * 56, max_pixels: int = 2048 * 2048, patch_size: int = 14, merge_size: int = 2,
                 temporal_patch_size: int = 2):
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.patch_size, self.merge_size, self.temporal_patch_size = patch_size, merge_size, temporal_patch_size
        # device: None = the reference's behaviour (fp32 tensors on the host); a cuda device = upload the resized uint8
        # image and do rescale / normalise / patch layout on the GPU, returning bf16 device tensors (bit-identical to the
        # host path followed by .to(bfloat16); set by vlm_fo1.model.builder for the MI355X engine)
        self.device = None
        self._lut = None

    def preprocess(self, images, videos=None, return_tensors: Optional[str] = "pt") -> Dict[str, torch.Tensor]:
        if videos is not None:
            raise NotImplementedError("video input is outside the hot path")
        if not isinstance(images, (list, tuple)):
            images = [images]
        p, m, t = self.patch_size, self.merge_size, self.temporal_patch_size
        pix, grids = [], []
        for img in images:
            img = img.convert("RGB")
            w, h = img.size
            rh, rw = smart_resize(h, w, p * m, self.min_pixels, self.max_pixels)
            if (rh, rw) != (h, w):
                img = img.resize((rw, rh), Image.Resampling.BICUBIC)
            gh, gw = rh // p, rw // p
            if self.device is not None:
                if t != 2:
                    raise NotImplementedError("device patchify is built for temporal_patch_size = 2")
                from vlm_fo1_amd import ops
                if self._lut is None:
                    self._lut = normalise_lut(OPENAI_CLIP_MEAN, OPENAI_CLIP_STD).to(self.device)
                u8 = torch.from_numpy(np.array(img, dtype=np.uint8)).to(self.device)   # blocking: the host array is a temporary
                pix.append(ops.patchify_u8(u8, self._lut, p, m))
                grids.append([1, gh, gw])
                continue
            a = _normalise(img, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)              # [3, rh, rw]
            x = np.broadcast_to(a[None], (t,) + a.shape)                          # frame duplicated along T
            x = x.reshape(t, 3, gh // m, m, p, gw // m, m, p)
            # -> (by, bx, dy, dx, C, T, py, px): rows in 2x2 merge-block order, vector (C, T, 14, 14)
            x = x.transpose(2, 5, 3, 6, 1, 0, 4, 7).reshape(gh * gw, 3 * t * p * p)
            pix.append(np.ascontiguousarray(x))
            grids.append([1, gh, gw])
        g = np.array(grids, dtype=np.int64)
        if self.device is not None:
            return {"pixel_values": pix[0] if len(pix) == 1 else torch.cat(pix, dim=0), "image_grid_thw": torch.from_numpy(g)}
        pv = np.concatenate(pix, axis=0)
        if return_tensors == "pt":
            return {"pixel_values": torch.from_numpy(pv), "image_grid_thw": torch.from_numpy(g)}
        return {"pixel_values": pv, "image_grid_thw": g}

    __call__ = preprocess


class CLIPStyleAuxProcessor:
    # image_processing_clip.py:98: the reference's default candidate sizes of the `dynamic_square` mode
    CANDIDATE_SIZES = (384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1280, 1536, 1792, 2048)

    def __init__(self, size: int = 768, resize_mode: str = "squash", image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD,
                 candidate_sizes: Optional[Sequence[int]] = None):
        self.size = size
        self.resize_mode = resize_mode
        self.do_resize = resize_mode != "dynamic"      # davit_aux_encoder.py:47-49
        self.candidate_sizes = tuple(candidate_sizes) if candidate_sizes else self.CANDIDATE_SIZES
        self.image_mean, self.image_std = image_mean, image_std
        self.device = None       # see Qwen2VLPatchProcessor.device
        self._lut = None
        if resize_mode not in ("squash", "dynamic", "dynamic_square"):
            raise NotImplementedError(f"aux resize mode {resize_mode!r} is not built (squash / dynamic / dynamic_square)")

    def target_hw(self, w: int, h: int):
        """Resized (width, height) of a w x h image.  squash: size x size.  dynamic_square (image_processing_clip.py:190-204): the square
        whose AREA is closest to the image's, among the candidate sizes (first one wins a tie: the reference compares with `<`)."""
        if self.resize_mode == "dynamic_square":
            area = w * h
            best = min(self.candidate_sizes, key=lambda c: abs(c * c - area))      # min() keeps the first of equal keys
            return best, best
        return self.size, self.size

    def preprocess(self, images, return_tensors: Optional[str] = "pt") -> Dict[str, object]:
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = []
        for img in images:
            img = img.convert("RGB")
            if self.do_resize:
                img = img.resize(self.target_hw(*img.size), Image.Resampling.BICUBIC)
            if self.device is not None:
                from vlm_fo1_amd import ops
                if self._lut is None:
                    self._lut = normalise_lut(self.image_mean, self.image_std).to(self.device)
                u8 = torch.from_numpy(np.array(img, dtype=np.uint8)).to(self.device)   # blocking: the host array is a temporary
                out.append(ops.normalize_u8(u8, self._lut))
                continue
            out.append(_normalise(img, self.image_mean, self.image_std))
        if self.device is not None:
            same = all(o.shape == out[0].shape for o in out)
            return {"pixel_values": torch.stack(out) if same else out}
        if return_tensors == "pt":
            same = all(o.shape == out[0].shape for o in out)
            return {"pixel_values": torch.from_numpy(np.stack(out)) if same else [torch.from_numpy(o) for o in out]}
        return {"pixel_values": out}

    __call__ = preprocess
