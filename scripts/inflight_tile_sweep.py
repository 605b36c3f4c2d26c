"""Does the best GEMM tile change when several images are in flight (occupancy gaps filled by other streams)?
Forces one tile shape / split-K setting for ALL GEMMs and measures 3-in-flight throughput.  GPU box only."""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vlm_fo1_amd import lib as L
dev = torch.device("cuda", 0)
case = bench.build_workload(dev)
R = 3
for label, tile, splitk in (("auto", 0, 0), ("all 128x128", 1, 0), ("all 64x128", 2, 0), ("all 64x64", 3, 0), ("auto tiles, no split-K", 0, 1)):
    L.load().fo1_gemm_set_variant(0, tile)
    L.load().fo1_gemm_set_splitk(splitk)
    pipe = bench.Pipeline(case, dev, inflight=R)
    for i in range(R):
        for _ in range(3):
            pipe.step(True, i)
    torch.cuda.synchronize()
    res = []
    for r in (1, R):
        K = 60
        t0 = time.perf_counter()
        for k in range(K):
            pipe.step(True, k % r)
        torch.cuda.synchronize()
        res.append(K / (time.perf_counter() - t0))
    print(f"{label:26s}: 1 in flight {res[0]:6.2f} images/s | {R} in flight {res[1]:6.2f} images/s", flush=True)
    del pipe
    torch.cuda.empty_cache()
L.load().fo1_gemm_set_variant(0, 0); L.load().fo1_gemm_set_splitk(0)
