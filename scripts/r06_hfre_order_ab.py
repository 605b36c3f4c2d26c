"""(round 6: + the work-list ORDER — box-major buckets (image-major walk, default) against the interleaved buckets of rounds 2-5 — at 1 / 12 / 25 images)
HFRE three-kernel form, per-kernel times (library event pairs) at 1 / 8 / 12 images x 100 boxes: scalar finish (round-2 first form)
vs the 16-byte finish; outputs compared bitwise.  usage: hfre_ab.py [out.json]"""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hfre_sweep_build import build                              # noqa: E402
from vlm_fo1_amd import lib as L                                # noqa: E402

lib = L.load()
res = []
for B in (1, 12, 25):
    m, call, out = build(B)
    m.worklist = True
    outs = {}
    for name, unroll, order in (("box_major_order", 8, -4), ("interleaved_order_r02", 8, -3), ("box_major_order_again", 8, -4)):
        L.check(lib.fo1_hfre_set_tuning(unroll, 512, 256, 4096), "set_tuning")
        L.check(lib.fo1_hfre_set_tuning(8, 512, order, 0), "order")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        L.profile(True)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        rows = L.profile_rows(reset=True)
        L.profile(False)
        outs[name] = out.clone()
        per = {r["name"]: round(r["total_ms"] / r["calls"] * 1e3, 2) for r in rows}
        tot = round(sum(per.values()), 2)
        bytes_ = [r for r in rows if r["name"] == "hfre_pool_items"][0]["total_work"] / 20
        res.append(dict(B=B, cfg=name, us=per, us_total=tot, us_per_image=round(tot / B, 2), full_map_bytes=bytes_,
                        gbps_full_map=round(bytes_ / tot / 1e3, 1)))
        print(res[-1], flush=True)
    same = bool(torch.equal(outs['box_major_order'], outs['interleaved_order_r02']))
    print(f"B={B}: box-major order == interleaved order bitwise: {same}", flush=True)
    res.append(dict(B=B, bitwise_box_major_equals_interleaved=same))
lib.fo1_hfre_set_tuning(8, 512, 256, 4096)
lib.fo1_hfre_set_tuning(8, 512, -4, 0)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
