import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hfre_cases import make_case
from test_hfre_gpu import to_dev, engine_out
from vlm_fo1_amd import lib as L
from vlm_fo1_amd import ops
lib = L.load()
d = to_dev(make_case("countbench30_fpn"))
ref = engine_out(d, worklist=True)
def ws_counter():
    torch.cuda.synchronize()
    out = {}
    for k, v in ops._ws_pool.items():
        if k[0] == "hfre_ex":
            out[k[3]] = int(v[:4].view(torch.int32)[0])
    return out
print("after default call: counters", ws_counter())
for (u, c, g) in [(8, 512, 2048), (8, 512, 64), (8, 512, 7), (8, 256, 2048), (8, 256, 64), (16, 128, 2048), (16, 128, 7), (8, 512, 2048)]:
    L.check(lib.fo1_hfre_set_tuning(u, c, 0, g), "t")
    got = engine_out(d, worklist=True)
    print((u, c, g), "equal", bool(torch.equal(got, ref)), "maxdiff", float((got - ref).abs().max()), "counters", ws_counter())
lib.fo1_hfre_set_tuning(8, 512, 0, 2048)
# repeated calls, then graph replays: does the counter stay put?
from vlm_fo1_amd.hfre import HFREModule
for i in range(3):
    engine_out(d, worklist=True)
    print("eager repeat", i, ws_counter())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o = engine_out(d, worklist=True)
for i in range(3):
    g.replay()
    print("graph replay", i, ws_counter(), "equal", bool(torch.equal(o, ref)))
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
bad = 0
for i in range(50):
    with torch.cuda.stream(side):
        for _ in range(3):
            a @ a
    g.replay()
    torch.cuda.synchronize()
    bad += int(not torch.equal(o, ref))
print("under load: differs", bad, "of 50", ws_counter())
