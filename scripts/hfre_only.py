"""Runs only the HFRE region pooling on the bench workload (for rocprofv3 --pmc passes and timing)."""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hfre_cases import box_fixtures, pyramid_sizes
from vlm_fo1_amd.hfre import HFREModule
n_boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = 480, 640
g = torch.Generator().manual_seed(1)
aux = [torch.randn(h * w, c, generator=g).bfloat16().cuda().view(1, h, w, c).permute(0, 3, 1, 2) for (h, w), c in zip(pyramid_sizes(H, W), (256, 512, 1024, 2048))]
gh, gw = 34, 46
fpn = [torch.randn(int(gh * f) * int(gw * f), 512, generator=g).bfloat16().cuda().view(1, int(gh * f), int(gw * f), 512).permute(0, 3, 1, 2) for f in (4, 2, 1, 0.5)]
it = [x for x in box_fixtures()["countbench" if n_boxes <= 32 else "pixmo"] if len(x["bboxes"]) >= n_boxes][0]
b = torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes] * torch.tensor([W / it["extent"][0], H / it["extent"][1]] * 2)
b = b.cuda()
m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, use_vision_tower_region_feature=True,
               vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True, simple_fpn=lambda x: fpn)
vt = torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
e1.record(); torch.cuda.synchronize()
maps_bytes = sum(t.numel() * 2 for t in aux + fpn)
print(f"hfre n_boxes={n_boxes}: {e0.elapsed_time(e1)/iters*1e3:.1f} us per call (3 kernels); maps {maps_bytes/1e6:.1f} MB, out {n_boxes*5888*4/1e6:.2f} MB")
from vlm_fo1_amd import lib as L
for budget in (64, 128, 256, 512, 1024, 4096):
    L.load().fo1_hfre_set_pixel_budget(budget)
    m._ws = None
    for _ in range(3):
        m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
    e1.record(); torch.cuda.synchronize()
    print(f"  pixel_budget={budget:5d}: {e0.elapsed_time(e1)/iters*1e3:.1f} us per call")
L.load().fo1_hfre_set_pixel_budget(0)
