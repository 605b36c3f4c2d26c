import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
dev = torch.device("cuda", 0)
case = bench.build_workload(dev)
pipe = bench.Pipeline(case, dev, inflight=1)
for _ in range(3): pipe.step(True)
torch.cuda.synchronize()
ent = list(pipe.eng._graphs.values())[0]
g = ent[0]
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
ts.sort()
print("hipGraphLaunch host time (median): %.2f ms; launch+execute: %.2f ms" % (ts[10][0] * 1e3, sorted(t[1] for t in ts)[10] * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    pipe.eng.llm.plan_inputs(case["ids"], 391, 32, (17, 23))
print("plan_inputs host: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
