"""Seeded HFRE test cases shared by the golden generator, the CPU tests and the GPU
parity tests.  Inputs are regenerated from seeds (bf16-valued maps); the golden file
stores a checksum of them so RNG drift is detected rather than silently compared."""
import hashlib
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO_BOXES = [[161.0, 11.0, 292.0, 127.0], [268.0, 61.0, 428.0, 226.0], [12.0, 100.0, 140.0, 227.0],
              [205.0, 188.0, 332.0, 320.0], [326.0, 202.0, 478.0, 357.0], [136.0, 106.0, 269.0, 233.0],
              [25.0, 206.0, 200.0, 383.0]]  # reference inference.py:16
AUX_DIMS = (256, 512, 1024, 2048)


def box_fixtures():
    return json.load(open(os.path.join(HERE, "box_fixtures.json")))


def pyramid_sizes(h, w):
    """DaViT stage sizes: conv7x7 s4 p3 then three conv3x3 s2 p1."""
    out = []
    h, w = (h + 3) // 4, (w + 3) // 4
    out.append((h, w))
    for _ in range(3):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    return out


def smart_grid(h, w):
    """ViT patch grid for an image (round to multiples of 28, no min/max clamp here)."""
    gh = max(2, round(h / 28) * 2)
    gw = max(2, round(w / 28) * 2)
    return gh, gw


CASES = {
    # name: (img_h, img_w, boxes-spec, fpn, seed)
    "demo_nofpn": dict(img=(399, 500), boxes="demo", fpn=False, seed=11),
    "demo_fpn": dict(img=(399, 500), boxes="demo", fpn=True, seed=12),
    "countbench30_fpn": dict(img=(346, 346), boxes=("countbench", 27), fpn=True, seed=13),
    "edge_fpn": dict(img=(240, 320), boxes="edge", fpn=True, seed=14),
    "pixmo100_small_nofpn": dict(img=(299, 376), boxes=("pixmo", 15), fpn=False, seed=15),
}


def make_boxes(spec, img_hw):
    H, W = img_hw
    if spec == "demo":
        return torch.tensor(DEMO_BOXES, dtype=torch.float32)
    if spec == "edge":
        # degenerate / border / full-image / sub-pixel / inverted-free cases, plus the
        # reference's dummy box for "no boxes" (omchat_qwen2_5_vl.py:90-91): [0,10,0,10]
        return torch.tensor([[0, 0, W, H], [0, 0, 1, 1], [W - 1, H - 1, W, H], [10, 10, 10, 10],
                             [W / 2, 0, W / 2 + 0.5, H], [0, H / 2, W, H / 2 + 0.25], [0, 10, 0, 10],
                             [3.3, 7.7, 44.4, 18.8], [W - 30, 5, W, 60], [5, H - 9, 90, H]], dtype=torch.float32)
    name, idx = spec
    it = [x for x in box_fixtures()[name] if x["index"] == idx][0]
    b = torch.tensor(it["bboxes"], dtype=torch.float32)
    ex, ey = it["extent"]
    b = b * torch.tensor([W / ex, H / ey, W / ex, H / ey])  # same rescale adjust_bbox does (mm_utils.py:296-311)
    return b


def make_case(name, dims_scale=1):
    """Returns dict(aux_maps[4x[1,C,H,W] bf16 token-major views], vt_maps (4 captured ViT maps
    [1,1280,gh,gw]) or fpn_maps (4x[1,512,..]), boxes (aux px), vt_boxes, grid_hw, region_dim)."""
    c = CASES[name]
    g = torch.Generator().manual_seed(c["seed"])
    H, W = c["img"]
    sizes = pyramid_sizes(H, W)
    aux = []
    for (h, w), ch in zip(sizes, AUX_DIMS):
        t = torch.randn(h * w, ch // dims_scale, generator=g).to(torch.bfloat16)
        aux.append(t.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0))
    gh, gw = smart_grid(H, W)
    out = dict(aux_maps=aux, grid_hw=(gh, gw), fpn=c["fpn"])
    if c["fpn"]:
        maps = []
        for f in (4, 2, 1, 0.5):
            h, w = int(gh * f), int(gw * f)
            t = torch.randn(h * w, 512 // dims_scale, generator=g).to(torch.bfloat16)
            maps.append(t.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0))
        out["fpn_maps"] = maps
        vt_c = 4 * (512 // dims_scale)
    else:
        maps = []
        for _ in range(4):
            t = torch.randn(gh * gw, 1280 // dims_scale, generator=g).to(torch.bfloat16)
            maps.append(t.reshape(gh, gw, -1).permute(2, 0, 1).unsqueeze(0))
        out["vt_maps"] = maps
        vt_c = 4 * (1280 // dims_scale)
    boxes = make_boxes(c["boxes"], (H, W))
    # encode_regions (omchat_qwen2_5_vl.py:94-99): vt box = aux box * (vt_size / aux_size), fp32
    sh = (gh * 14) / H
    sw = (gw * 14) / W
    out["boxes"] = boxes
    out["vt_scale"] = (sw, sh)
    out["vt_boxes"] = boxes * torch.tensor([sw, sh, sw, sh])
    out["region_dim"] = sum(AUX_DIMS) // dims_scale + vt_c
    return out


def checksum(case):
    h = hashlib.sha256()
    for k in ("aux_maps", "fpn_maps", "vt_maps"):
        for t in case.get(k, []):
            h.update(t.permute(0, 2, 3, 1).contiguous().view(torch.int16).numpy().tobytes())
    h.update(case["boxes"].numpy().tobytes())
    return h.hexdigest()
