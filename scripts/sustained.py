"""Sustained-load check (GPU box only): the default bench pipeline for N steps, wall time per block of 10 steps, plus rocm-smi clocks /
power before and after.  usage: sustained.py [steps] [batch] [inflight]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def smi(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "junction", "edge"))]
        print(tag, " | ".join(keep)[:600], flush=True)
    except Exception as e:   # noqa: BLE001
        print(tag, "rocm-smi failed:", e)


dev = torch.device("cuda", 0)
cases = [bench.build_workload(dev, n_boxes=100, seed=1234 + i) for i in range(B)]
pipe = bench.Pipeline(cases[0], dev, inflight=R, batch=B, cases=cases)
for s in range(R):
    for _ in range(4):
        pipe.step(True, s)
torch.cuda.synchronize()
if len(sys.argv) > 4 and sys.argv[4] == "nogc":
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    print("cyclic GC parked (gc.freeze + gc.disable)")
smi("before:")
t0 = time.perf_counter()
last = t0
for k in range(steps):
    pipe.step(True, k % R)
    if (k + 1) % 10 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        print(f"steps {k - 8:4d}-{k + 1:4d}: {(now - last) / 10 * 1e3:7.1f} ms/step  {10 * B / (now - last):6.1f} images/s   t = {now - t0:5.1f} s", flush=True)
        if (k + 1) % 50 == 0:
            smi("   smi:")
        last = time.perf_counter()
smi("after:")
