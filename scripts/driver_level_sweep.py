#!/usr/bin/env python3
"""bench.py's driver_level block (evaluation/eval_coco.py's own loop) over worker / prefetch-thread / pass-size settings (GPU box only):
picks the drivers' defaults ($FO1_BATCH, $FO1_INFLIGHT, $FO1_PREFETCH_THREADS).  usage: driver_level_sweep.py <out.json> [items]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.set_num_threads(1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 768
cases = [B.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(2)]
pipe = B.Pipeline(cases[0], dev, inflight=1, batch=2, cases=cases)
res = []
for inflight, threads, batch, pool in ((2, 4, 32, 128), (3, 4, 32, 128), (4, 4, 32, 128), (2, 2, 32, 128), (2, 4, 32, 128)):
    r = B.driver_level_run(pipe, n_items=n, n_warm=max(128, inflight * batch), K=64, batch=batch, inflight=inflight, pool_slots=pool, prefetch_threads=threads)
    pipe.eng.__dict__.pop("_unused", None)
    res.append(r)
    print(json.dumps({k: r[k] for k in ("images_per_sec", "items", "seconds", "images_per_pass", "prefill_workers", "decode_pool_slots", "prefetch_threads")}), flush=True)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/driver_level_sweep.json", "w"), indent=1)
