#!/usr/bin/env python3
"""Round 6 experiment: ONE decode pool of 128 slots against TWO pools of 64 stepping concurrently on two HIP streams (each kernel of a step is
a short dependent launch with a ramp and a tail; two chains interleave on the GPU, and the second reader of a weight finds it in the 256 MB
Infinity Cache while the chains stay within a layer of each other).  Also: gate/up on the <128, 96> deep-ring tile (FO1_AB pin) for the pool
of 128.  GPU box only.  usage: r06_two_pools.py <out.json>"""
import os
os.environ.setdefault("FO1_AB", "1")
import json
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vlm_fo1_amd import lib as L
from vlm_fo1_amd.llm import DecodePool

dev = torch.device("cuda", 0)
cases = [bench.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(32)]
pipe = bench.Pipeline(cases[0], dev, inflight=1, batch=32, cases=cases)
eng = pipe.eng
reqs = pipe.requests[:32]
eng.prefill_batch(reqs, use_graph=False)
torch.cuda.synchronize()
hp, first = eng._last_batch, eng._last_next_tokens.clone()
STEPS = 48
res = {}


def fill(pool):
    left = pool.P
    while left > 0:
        n = min(len(reqs), left)
        pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][:n], hp["delta"][:n], first[:n], 300, ())
        left -= n


def one(slots, tag):
    pool = DecodePool(eng.llm, slots=slots)
    fill(pool)
    for _ in range(4):
        pool.step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        pool.step(True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / STEPS
    res[tag] = dict(ms_per_step=round(t * 1e3, 3), tokens_per_sec=round(slots / t, 1))
    print(tag, res[tag], flush=True)
    del pool
    torch.cuda.empty_cache()


def two(slots, tag):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    pools = []
    for s in streams:
        with torch.cuda.stream(s):
            p = DecodePool(eng.llm, slots=slots)
            fill(p)
            for _ in range(4):
                p.step(True)
            pools.append(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for p, s in zip(pools, streams):
            with torch.cuda.stream(s):
                p.step(True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / STEPS
    res[tag] = dict(ms_per_step_pair=round(t * 1e3, 3), tokens_per_sec=round(2 * slots / t, 1))
    print(tag, res[tag], flush=True)
    del pools
    torch.cuda.empty_cache()


one(128, "one_pool_128")
one(64, "one_pool_64")
two(64, "two_pools_64_concurrent")
two(128, "two_pools_128_concurrent")
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_two_pools.json", "w"), indent=1)
