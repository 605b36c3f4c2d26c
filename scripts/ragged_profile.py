"""Per-pass anatomy of a dataset-shaped (ragged) run: host enqueue time vs GPU time per packed pass, for several packing policies.
    python scripts/ragged_profile.py [countbench|pixmo|coco-like] [items]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import bench_workloads as BW  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "countbench"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    dev = torch.device("cuda", 0)
    case = bench.build_workload(dev, n_boxes=100)
    pipe = bench.Pipeline(case, dev, inflight=2, batch=1, cases=[case])
    reqs, geos = BW.build_requests(name, dev, limit=n)
    out = {}
    for label, batch, budget in (("25 images / 58k rows", 25, 25 * 1564 * 3 // 2), ("64 images / 40k rows", 64, 40000), ("128 images / 60k rows", 128, 60000)):
        groups = BW.pack(geos, batch=batch, row_budget=budget)
        need = max(sum(len(reqs[i]["ids"]) + geos[i]["S"] // 4 + 8 for i in g) for g in groups)
        for e in pipe.engs:
            e.llm.reserve(need)
        rows = []
        for sweep in range(2):
            rows = []
            for g in groups:
                grp = [reqs[i] for i in g]
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                pipe.eng.prefill_batch(grp, use_graph=False)
                e1.record()
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                rows.append(dict(images=len(g), vit_rows=sum(geos[i]["S"] for i in g), boxes=sum(geos[i]["n"] for i in g),
                                 host_ms=round(t_host * 1e3, 1), gpu_ms=round(e0.elapsed_time(e1), 1)))
        tot_gpu = sum(r["gpu_ms"] for r in rows)
        tot_host = sum(r["host_ms"] for r in rows)
        patches = sum(g["S"] for g in geos)
        # two in flight, timed sweep
        def sweep2():
            for k, g in enumerate(groups):
                with torch.cuda.stream(pipe.streams[k % 2]):
                    pipe.engs[k % 2].prefill_batch([reqs[i] for i in g], use_graph=False)
            torch.cuda.synchronize()
        sweep2()
        t0 = time.perf_counter()
        sweep2()
        el = time.perf_counter() - t0
        out[label] = dict(passes=len(groups), sum_gpu_ms=round(tot_gpu, 1), sum_host_ms=round(tot_host, 1),
                          uniform_equivalent_img_per_s_serial_gpu=round(patches / 1564 / (tot_gpu * 1e-3), 1),
                          two_in_flight_images_per_sec=round(len(reqs) / el, 1), two_in_flight_uniform_equivalent=round(patches / 1564 / el, 1), per_pass=rows)
        print(label, json.dumps({k: v for k, v in out[label].items() if k != "per_pass"}))
        for r in rows:
            print("   ", r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"ragged_profile_{name}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
