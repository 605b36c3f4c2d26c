#!/usr/bin/env python3
"""Where the HOST threads of the driver-level loop spend their wall time (GPU box only): wall-clock wrappers around the host-side
sections of evaluation/eval_coco.py's loop (prepare_inputs on the prefetch threads; request building, host planning + launches of the
packed prefill, hand-over to the decode pool, waiting for the ids on the worker threads; join / step / harvest on the pool thread).
usage: driver_level_hostprofile.py [items] [inflight] [prefetch_threads]"""
import collections
import json
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "evaluation"))
import torch
import bench as B
import eval_coco as E
from vlm_fo1.model.fo1_model import FO1ForCausalLM
from vlm_fo1_amd import llm as LLM, model as M, serving as S, sharded_eval as SE

acc = collections.defaultdict(lambda: [0, 0.0])
lock = threading.Lock()


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            key = (threading.current_thread().name.split("_")[0].rstrip("0123456789-"), label)
            with lock:
                acc[key][0] += 1
                acc[key][1] += dt
    setattr(obj, name, w)


for obj, name in ((E, "prepare_inputs"), (FO1ForCausalLM, "generate_many"), (FO1ForCausalLM, "_request"), (M.FO1Engine, "prefill_batch"), (M.FO1Engine, "submit_batch"),
                  (LLM.QwenLLM, "plan_batch"), (S.PoolService, "submit"), (S.PoolHandle, "wait_relocated"), (S.PoolHandle, "result"),
                  (LLM.DecodePool, "join"), (LLM.DecodePool, "step"), (LLM.DecodePool, "harvest"), (LLM.DecodePool, "snapshot"), (SE.Prefetcher, "get")):
    wrap(obj, name)

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 4
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cases = [B.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(2)]
pipe = B.Pipeline(cases[0], dev, inflight=1, batch=2, cases=cases)
B.driver_level_run(pipe, n_items=128, n_warm=128, K=64, batch=32, inflight=inflight, pool_slots=128, prefetch_threads=threads)    # replicas, graphs
with lock:
    acc.clear()
r = B.driver_level_run(pipe, n_items=n, n_warm=128, K=64, batch=32, inflight=inflight, pool_slots=128, prefetch_threads=threads)
print(json.dumps({k: r[k] for k in ("images_per_sec", "items", "seconds", "prefill_workers", "prefetch_threads")}))
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
print(f"{'thread':16s} {'section':34s} {'calls':>7s} {'total s':>9s} {'ms/call':>9s}   (both eval_coco calls of driver_level_run: warm-up 128 + {n} items)")
for (th, label), (c, t) in rows:
    print(f"{th:16s} {label:34s} {c:7d} {t:9.3f} {t / c * 1e3:9.3f}")
