#!/usr/bin/env python3
"""BASELINE configs[4]'s geometry (1344x1344 x 300 proposals = 3 prompts of 100 over one image), shared-prefix LLM prefill on / off
(FO1Engine.SHARE_PREFIX), bf16 and fp8 linears (GPU box only).  usage: hires_ab.py [out.json]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from vlm_fo1_amd.model import FO1Engine

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.set_num_threads(1)
cases = [B.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(2)]
pipe = B.Pipeline(cases[0], dev, inflight=2, batch=2, cases=cases)
out = {}
for share in (False, True):
    FO1Engine.SHARE_PREFIX = share
    for e in pipe.engs:
        e._graphs.clear(); e._seen.clear()
    r = B.hires_run(pipe)
    out["shared_prefix" if share else "full_rows"] = r
    print(share, json.dumps({k: r[k] for k in ("bf16", "fp8")}), flush=True)
FO1Engine.SHARE_PREFIX = True
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
