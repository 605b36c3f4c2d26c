"""Per-kernel time of one batched decode step (GPU box only): eager launches with the library's per-launch event pairs.
usage: decode_batch_profile.py [B] [kv_len]"""
import os
os.environ.setdefault("FO1_AB", "1")   # A/B switches live in the test / bench build only (include/fo1_ab.h)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from vlm_fo1_amd import lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if "FO1_GEMV_RPL" in os.environ:
    L.check(L.load().fo1_gemv_batch_set_rows_per_lane(int(os.environ["FO1_GEMV_RPL"])), "set_rpl")
    print("rows per lane setting =", os.environ["FO1_GEMV_RPL"])
dev = torch.device("cuda", 0)
cases = [bench.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(B)]
pipe = bench.Pipeline(cases[0], dev, inflight=1, batch=B, cases=cases)
eng = pipe.eng
eng.prefill_batch(pipe.requests, use_graph=False) if B > 1 else pipe.step_single(False)
if B == 1:
    eng.prefill_batch(pipe.requests[:1], use_graph=False)
d = eng._decoder()
hp = eng._last_batch
d.start(hp["seqs"], hp["delta"], eng._last_next_tokens[:B], 4096, ())
for _ in range(3):
    d.step(False)
torch.cuda.synchronize()
L.load().fo1_gemm_profile_shapes(1)
L.profile(True)
for _ in range(5):
    d.step(False)
torch.cuda.synchronize()
rows = L.profile_rows(reset=True)
L.profile(False)
tot = sum(r["total_ms"] for r in rows) / 5
print(f"B={B}: kernel time per step {tot:.3f} ms")
for r in sorted(rows, key=lambda r: -r["total_ms"]):
    per = r["total_ms"] / r["calls"] * 1e3
    bw = r["total_work"] / r["calls"] / (per * 1e-6) / 1e9 if per > 0 else 0
    print(f"{r['name']:34s} x{r['calls'] // 5:4d}  {per:8.2f} us  {r['total_ms'] / 5:8.3f} ms/step  {bw:8.1f} GB/s(work)")
# graph replay time
for _ in range(3):
    d.step(True)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(32):
    d.step(True)
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t) / 32 * 1e3:.3f} ms/step")
