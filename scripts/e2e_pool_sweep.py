#!/usr/bin/env python3
"""bench.py's end_to_end loop (upload -> preprocessing -> packed prefill -> 64-token decode through the decode pool) over feeding replicas,
images per pass and pool steps per scheduler round (GPU box only).  usage: e2e_pool_sweep.py [out.json]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from vlm_fo1_amd import serving

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.set_num_threads(1)
cases = [B.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(32)]
res = []
for R, per_pass, spr in ((2, 32, 4), (3, 32, 4), (2, 25, 4), (2, 32, 2), (2, 32, 8), (3, 25, 4)):
    pipe = B.Pipeline(cases[0], dev, inflight=R, batch=per_pass, cases=cases[:per_pass])
    orig = serving.PoolService.__init__

    def patched(self, llm, slots=128, slot_rows=1024, steps_per_round=4, use_graph=True, _o=orig, _s=spr):
        _o(self, llm, slots=slots, slot_rows=slot_rows, steps_per_round=_s, use_graph=use_graph)
    serving.PoolService.__init__ = patched
    try:
        r = B.end_to_end_run(pipe, cases[:per_pass], steps=24, K=64, pool_slots=128)
    finally:
        serving.PoolService.__init__ = orig
    row = dict(replicas=R, images_per_pass=per_pass, steps_per_round=spr, images_per_sec=r["images_per_sec"], mean_live=r["pool_mean_live_sequences"])
    res.append(row)
    print(json.dumps(row), flush=True)
    del pipe
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
