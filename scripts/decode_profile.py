"""Per-kernel time of one eager decode step (hipEvent profile), bench workload."""
import sys, os
os.environ.setdefault("FO1_AB", "1")   # fo1_gemm_profile_shapes is an instrument of the test / bench build (include/fo1_ab.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from vlm_fo1_amd import lib as L
dev = torch.device("cuda", 0)
case = bench.build_workload(dev)
pipe = bench.Pipeline(case, dev)
out = pipe.step(False)
tok = out["next_token"]
llm = pipe.eng.llm
for _ in range(3):
    _, _, tok = llm.decode_step(tok)
torch.cuda.synchronize()
L.load().fo1_gemm_profile_shapes(1)
L.profile(True)
N = 10
for _ in range(N):
    _, _, tok = llm.decode_step(tok)
torch.cuda.synchronize()
rows = L.profile_rows()
L.profile(False)
rows.sort(key=lambda r: -r["total_ms"])
tot = 0
for r in rows:
    print(f"{r['name']:28s} {r['total_ms']/N*1e3:9.1f} us/step  x{r['calls']//N:4d}  avg {r['total_ms']/r['calls']*1e3:7.2f} us  {r['total_work']/r['total_ms']/1e6 if r['total_ms'] else 0:8.1f} GB/s(or GF/s)")
    tot += r["total_ms"] / N
print("sum", tot, "ms/step")
