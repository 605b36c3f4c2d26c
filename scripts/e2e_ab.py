"""End-to-end anatomy of one pass of 25 images (prefill + 64-token decode), one pass at a time: ONE decode group of 25 (round 3: two
16-column MFMA groups per weight fragment) vs round 2's groups of <= 16, sequential vs concurrent.   python scripts/e2e_ab.py [--more]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
cases = [bench.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(25)]
pipe = bench.Pipeline(cases[0], dev, inflight=1, batch=25, cases=cases)
eng = pipe.eng
K = 64


def run(label, concurrent, groups, reps=4, gmax=16):
    eng.DECODE_CONCURRENT, eng.DECODE_GROUPS, eng.DECODE_MAX_GROUP = concurrent, groups, gmax
    for _ in range(2):
        eng.generate_batch(pipe.requests, max_new_tokens=K, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.generate_batch(pipe.requests, max_new_tokens=K, use_graph=True)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print(f"{label:40s} {el * 1e3:8.1f} ms per pass of 25 -> {25 / el:6.1f} images/s", flush=True)
    return el


for _ in range(3):
    eng.prefill_batch(pipe.requests, use_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.prefill_batch(pipe.requests, use_graph=True)
torch.cuda.synchronize()
print(f"prefill only: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms", flush=True)
run("ONE group of 25 (two MFMA column groups)", True, 1, gmax=32)
run("2 groups (13 + 12) one after the other", False, 2)
run("2 groups (13 + 12) together", True, 2)
run("ONE group of 25 again", True, 1, gmax=32)
if "--more" in sys.argv:
    run("3 groups (9 + 8 + 8) together", True, 3)
    run("4 groups together", True, 4)
