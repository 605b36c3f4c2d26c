"""Kernel-only times (dispatch timestamps) of the three HFRE kernels on the bench geometry.  usage: hfre_kernel_times.py [n_boxes ...]"""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hfre_cases import box_fixtures, pyramid_sizes
from vlm_fo1_amd import lib as L
from vlm_fo1_amd.hfre import HFREModule
H, W = 480, 640
g = torch.Generator().manual_seed(1)
aux = [torch.randn(h * w, c, generator=g).bfloat16().cuda().view(1, h, w, c).permute(0, 3, 1, 2) for (h, w), c in zip(pyramid_sizes(H, W), (256, 512, 1024, 2048))]
gh, gw = 34, 46
fpn = [torch.randn(int(gh * f) * int(gw * f), 512, generator=g).bfloat16().cuda().view(1, int(gh * f), int(gw * f), 512).permute(0, 3, 1, 2) for f in (4, 2, 1, 0.5)]
vt = torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda")
for n_boxes in [int(a) for a in sys.argv[1:]] or [32, 100]:
    it = [x for x in box_fixtures()["countbench" if n_boxes <= 32 else "pixmo"] if len(x["bboxes"]) >= n_boxes][0]
    b = (torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes] * torch.tensor([W / it["extent"][0], H / it["extent"][1]] * 2)).cuda()
    m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, use_vision_tower_region_feature=True,
                   vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True, simple_fpn=lambda x: fpn)
    for budget in (0, 64, 128, 256, 512):
        L.load().fo1_hfre_set_pixel_budget(budget)
        for _ in range(3):
            m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
        torch.cuda.synchronize()
        L.profile(True)
        for _ in range(20):
            m(aux, [b], vt, None, vt_scale=(gw * 14 / W, gh * 14 / H))
        torch.cuda.synchronize()
        rows = L.profile_rows()
        L.profile(False)
        print(f"n_boxes={n_boxes} budget={budget or 'auto'}: " + ", ".join(f"{r['name']} {r['total_ms'] / r['calls'] * 1e3:.1f} us" for r in rows) +
              f"; sum {sum(r['total_ms'] / r['calls'] for r in rows) * 1e3:.1f} us", flush=True)
