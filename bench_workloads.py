"""Dataset-shaped synthetic workloads for bench.py and the GPU tests (BASELINE configs[2], configs[3]).

No image or checkpoint exists offline, so pixel content is seeded noise — but the GEOMETRY is the datasets' own:
  * `countbench` / `pixmo`: every item of the reference's evaluation fixtures with its UPN box list verbatim
    (vlm_fo1_amd/fixtures/dataset_boxes.npz: 487 items / 11 144 boxes and 529 items / 28 996 boxes, N in [2, 100]); the image is synthesised
    at the extent of its boxes, max(x2) x max(y2) — 99 x 99 up to 5181 x 3444 pixels;
  * `coco-like`: image sizes cycled over a COCO-val2017-like list (640x480 dominant, portrait and 4:3 / 3:2 variants), 100 boxes per
    image drawn (seed 1234) from the empirical normalised (x1, y1, x2, y2) distribution of the Pixmo fixture.
Sizing follows the product path item by item (vlm_fo1/mm_utils.py prepare_inputs): long side capped at 2048 and both sides >= 28
(`resize_shortest_edge_images_and_bboxes`), at most 100 boxes (`:600`), boxes clamped and rescaled into the aux tensor's pixel space
(`adjust_bbox`), primary grid from smart-resize to multiples of 28, aux image `dynamic` (= the resized image) or `squash` (768 x 768).
The prompt is the synthetic sentinel sequence of vlm_fo1_amd.model.synthetic_prompt (60 text ids + 2 per region)."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

# (width, height) x weight: the head of COCO val2017's size histogram (landscape 4:3 / 3:2 dominate), enough variety to break any
# "every image has the same geometry" assumption
COCO_SIZES = [((640, 480), 25), ((640, 427), 18), ((480, 640), 8), ((640, 426), 7), ((500, 375), 6), ((427, 640), 5), ((640, 428), 4),
              ((500, 333), 4), ((640, 425), 3), ((375, 500), 3), ((640, 360), 3), ((640, 512), 2), ((333, 500), 2), ((612, 612), 2),
              ((640, 478), 2), ((500, 400), 2), ((640, 457), 1), ((512, 640), 1), ((640, 640), 1), ((359, 640), 1)]


def _fixture():
    return np.load(os.path.join(ROOT, "vlm_fo1_amd", "fixtures", "dataset_boxes.npz"))


def dataset_items(name: str, limit: Optional[int] = None, seed: int = 1234) -> List[Dict]:
    """-> [dict(width, height, boxes float32 [N,4] in ORIGINAL image pixels)] in the dataset's own order."""
    fx = _fixture()
    items = []
    if name in ("countbench", "pixmo"):
        counts, boxes = fx[name + "_counts"].astype(np.int64), fx[name + "_boxes"].astype(np.float32)
        off = np.concatenate([[0], np.cumsum(counts)])
        for i in range(len(counts)):
            b = boxes[off[i]:off[i + 1]]
            items.append(dict(width=int(b[:, 2].max()), height=int(b[:, 3].max()), boxes=b))
    elif name == "coco-like":
        counts, boxes = fx["pixmo_counts"].astype(np.int64), fx["pixmo_boxes"].astype(np.float32)
        off = np.concatenate([[0], np.cumsum(counts)])
        pool = []
        for i in range(len(counts)):                       # normalised boxes of every Pixmo item
            b = boxes[off[i]:off[i + 1]]
            pool.append(b / np.array([b[:, 2].max(), b[:, 3].max()] * 2, dtype=np.float32))
        pool = np.concatenate(pool)
        rng = np.random.default_rng(seed)
        sizes = [s for s, w in COCO_SIZES for _ in range(w)]
        rng.shuffle(sizes)
        n = limit or 5000
        for i in range(n):
            w, h = sizes[i % len(sizes)]
            b = pool[rng.integers(0, len(pool), 100)] * np.array([w, h, w, h], dtype=np.float32)
            items.append(dict(width=w, height=h, boxes=np.round(b).astype(np.float32)))
    else:
        raise ValueError(f"unknown dataset {name!r} (countbench | pixmo | coco-like)")
    return items[:limit] if limit else items


def geometry(item: Dict, aux_mode: str = "dynamic") -> Dict:
    """The shapes prepare_inputs would produce for this item: resized image, patch grid, aux size, boxes in aux pixels."""
    from vlm_fo1.mm_utils import adjust_bbox
    from vlm_fo1.model.image_processing import smart_resize
    w0, h0 = item["width"], item["height"]
    h1, w1 = h0, w0
    if max(h1, w1) > 2048:                                  # resize_shortest_edge_images_and_bboxes(max_size=2048), mm_utils.py:229-258
        shrink = 2048 / max(h1, w1)
        h1, w1 = int(h1 * shrink), int(w1 * shrink)
    w1, h1 = max(28, w1), max(28, h1)
    rx, ry = w1 / w0, h1 / h0
    boxes = [[x1 * rx, y1 * ry, x2 * rx, y2 * ry] for x1, y1, x2, y2 in item["boxes"].tolist()][:100]      # :600
    rh, rw = smart_resize(h1, w1, 28, 56 * 56, 2048 * 2048)
    ah, aw = (768, 768) if aux_mode == "squash" else (h1, w1)
    boxes = torch.tensor(adjust_bbox(boxes, h1, w1, ah, aw), dtype=torch.float32)
    return dict(grid=(rh // 14, rw // 14), aux_hw=(ah, aw), boxes=boxes, n=boxes.shape[0], S=(rh // 14) * (rw // 14))


def build_requests(name: str, device, limit: Optional[int] = None, aux_mode: str = "dynamic", seed: int = 1234, vocab: int = 151936):
    """-> (requests for FO1Engine.prefill_batch / generate_batch with every tensor resident on `device`, per-item geometry dicts)."""
    from vlm_fo1_amd.model import synthetic_prompt
    items = dataset_items(name, limit, seed)
    g = torch.Generator(device=device).manual_seed(seed)
    reqs, geos = [], []
    for i, it in enumerate(items):
        geo = geometry(it, aux_mode)
        gh, gw = geo["grid"]
        pix = torch.randn(gh * gw, 1176, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        aux = torch.randn(3, *geo["aux_hw"], generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        reqs.append(dict(ids=synthetic_prompt(geo["n"], n_text=60, vocab=vocab, seed=seed + i), pix=pix, grid=(gh, gw), aux=aux, boxes=geo["boxes"].to(device)))
        geos.append(geo)
    return reqs, geos


def pack(geos: List[Dict], batch: int = 25, row_budget: int = 25 * 1564 * 3 // 2, sort: bool = True) -> List[List[int]]:
    """Passes of at most `batch` items and at most `row_budget` ViT rows (one giant image is its own pass); items of similar cost share a
    pass when `sort` (what sharded_eval.run_sharded does with its cost key)."""
    from vlm_fo1_amd.sharded_eval import item_cost
    order = list(range(len(geos)))
    if sort:
        order.sort(key=lambda i: (item_cost(geos[i]["aux_hw"][1], geos[i]["aux_hw"][0], geos[i]["n"]), i))
    groups, cur, rows = [], [], 0
    for i in order:
        s = geos[i]["S"]
        if cur and (len(cur) >= batch or rows + s > row_budget):
            groups.append(cur)
            cur, rows = [], 0
        cur.append(i)
        rows += s
    if cur:
        groups.append(cur)
    return groups


def summary(geos: List[Dict]) -> Dict:
    S = np.array([g["S"] for g in geos])
    n = np.array([g["n"] for g in geos])
    return dict(items=len(geos), boxes=int(n.sum()), boxes_per_item=round(float(n.mean()), 1), patches_per_item=round(float(S.mean()), 1),
                patches_min=int(S.min()), patches_max=int(S.max()), distinct_grids=len({g["grid"] for g in geos}),
                distinct_aux_sizes=len({g["aux_hw"] for g in geos}))
