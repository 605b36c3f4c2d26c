"""Throughput with several single-image passes in flight: R engine replicas (shared weights), each replaying its own hipGraph
on its own stream — submitted round-robin from ONE host thread, or from one host thread PER replica (hipGraphLaunch costs
~12 us of host time per node, ~12 ms for the ~950-node pass, so a single submitting thread saturates near 80 images/s).
usage: inflight_bench.py [R ...]   (GPU box only)"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
Rs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
case = bench.build_workload(dev)
pipe = bench.Pipeline(case, dev, inflight=max(Rs))
for i in range(max(Rs)):
    for _ in range(3):
        pipe.step(True, i)
torch.cuda.synchronize()
K = 120
for R in Rs:
    t0 = time.perf_counter()
    for k in range(K):
        pipe.step(True, k % R)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    line = f"in flight {R}: one submitting thread {K / el:7.2f} images/s ({el / K * 1e3:.2f} ms per image)"
    if R > 1:
        start = threading.Barrier(R + 1)

        def loop(i):
            torch.cuda.set_device(0)
            start.wait()
            for _ in range(K // R):
                pipe.step(True, i)
            pipe.streams[i].synchronize()

        th = [threading.Thread(target=loop, args=(i,)) for i in range(R)]
        [t.start() for t in th]
        start.wait()
        t0 = time.perf_counter()
        [t.join() for t in th]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n = (K // R) * R
        line += f" | one thread per replica {n / el:7.2f} images/s ({el / n * 1e3:.2f} ms per image)"
    print(line, flush=True)
