"""Throughput with several single-image passes in flight: R engine replicas, each replaying its own hipGraph on its own
stream (round-robin submission from one host thread).  usage: inflight_bench.py [R ...]   (GPU box only)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
Rs = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
pipes, streams = [], []
for R in Rs:
    while len(pipes) < R:
        case = bench.build_workload(dev, seed=1234 + len(pipes))
        pipes.append(bench.Pipeline(case, dev))
        streams.append(torch.cuda.Stream())
    for i in range(R):
        with torch.cuda.stream(streams[i]):
            for _ in range(3):
                pipes[i].step(True)
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for k in range(K):
        i = k % R
        with torch.cuda.stream(streams[i]):
            pipes[i].step(True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"in flight {R}: {K / el:7.2f} images/s  ({el / K * 1e3:.2f} ms per image)", flush=True)
