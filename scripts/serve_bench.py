"""Requests per second INCLUDING a greedy answer of K tokens (default 64): R worker threads, each with its own engine replica
and HIP stream (vlm_fo1_amd.sharded_eval.request_workers-style), configs[1] geometry.  usage: serve_bench.py [K] [R ...]"""
import sys, os, time, threading, queue
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Rs = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
case = bench.build_workload(dev)
pipe = bench.Pipeline(case, dev, inflight=max(Rs))
d = case["dev"]
N = 24


def one(eng):
    return eng.generate(case["ids"], d["pix"], case["grid"], d["aux"], d["boxes"], max_new_tokens=K, use_graph=True)


for R in Rs:
    engs, streams = pipe.engs[:R], [torch.cuda.Stream() for _ in range(R)]
    for e, s in zip(engs, streams):      # warm-up: graph captures
        with torch.cuda.stream(s):
            ref = one(e)
    torch.cuda.synchronize()
    q = queue.Queue()
    for i in range(N):
        q.put(i)
    outs = []

    def loop(e, s):
        torch.cuda.set_device(0)
        while True:
            try:
                q.get_nowait()
            except queue.Empty:
                return
            with torch.cuda.stream(s):
                outs.append(one(e))

    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(e, s)) for e, s in zip(engs, streams)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert all(o == ref for o in outs), "answers differ between workers"
    print(f"requests in flight {R}: {N / el:6.2f} images/s with a {K}-token answer ({el / N * 1e3:.1f} ms per image; {N * K / el:.0f} tokens/s)", flush=True)
