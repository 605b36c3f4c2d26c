#!/usr/bin/env python3
"""bench.py — throughput of the VLM-FO1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one image through every hot-path stage the engine implements (listed in
config.stages), inputs already resident in HBM.  N>1: images shard across ranks with no
data-path collective (weak scaling); value = images all ranks processed / max-over-ranks
time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16


def build_workload(device, n_boxes=32, img_hw=(480, 640), seed=1234):
    """BASELINE.json configs[1]: 1 image (640x480 synthetic) x 32 proposals (first 32 boxes of the
    CountBench fixture item with N>=32, rescaled to the image), true channel counts."""
    from hfre_cases import box_fixtures, pyramid_sizes
    H, W = img_hw
    g = torch.Generator().manual_seed(seed)
    sizes = pyramid_sizes(H, W)
    aux = [torch.randn(h * w, c, generator=g).bfloat16().reshape(h, w, c).permute(2, 0, 1).unsqueeze(0)
           for (h, w), c in zip(sizes, (256, 512, 1024, 2048))]
    gh, gw = round(H / 28) * 2, round(W / 28) * 2
    fpn = []
    for f in (4, 2, 1, 0.5):
        h, w = int(gh * f), int(gw * f)
        fpn.append(torch.randn(h * w, 512, generator=g).bfloat16().reshape(h, w, 512).permute(2, 0, 1).unsqueeze(0))
    items = [x for x in box_fixtures()["countbench"] if len(x["bboxes"]) >= n_boxes]
    it = items[0]
    b = torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes]
    ex, ey = it["extent"]
    b = b * torch.tensor([W / ex, H / ey, W / ex, H / ey])
    sw, sh = gw * 14 / W, gh * 14 / H
    case = dict(aux_maps=aux, fpn_maps=fpn, fpn=True, grid_hw=(gh, gw), boxes=b, vt_scale=(sw, sh),
                vt_boxes=b * torch.tensor([sw, sh, sw, sh]), region_dim=5888, img_hw=img_hw)
    if device is not None:
        case["dev"] = dict(
            aux_maps=[m.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2) for m in aux],
            fpn_maps=[m.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2) for m in fpn],
            boxes=b.to(device), vt_boxes=case["vt_boxes"].to(device),
            vt_in=torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device=device))
    return case


class Pipeline:
    """The stages of the hot path implemented so far, run back to back on one stream."""
    stages = ["hfre_region_pool"]

    def __init__(self, case):
        from vlm_fo1_amd.hfre import HFREModule
        self.d = case["dev"]
        self.hfre = HFREModule(roi_output_size=7, region_feature_dim=case["region_dim"], apply_position_embedding=True,
                               use_vision_tower_region_feature=True, vision_tower_region_feature_dim=2048,
                               use_simpleFPN_for_vt=True, simple_fpn=lambda x: self.d["fpn_maps"])

    def step(self):
        d = self.d
        return self.hfre(d["aux_maps"], [d["boxes"]], d["vt_in"], [d["vt_boxes"]])


def cpu_baseline(case, budget_s=15.0):
    """Oracle (port) of the same stages on the host cores, bounded sample."""
    from oracle import hfre_oracle as O
    torch.set_num_threads(os.cpu_count())
    n = 0
    t0 = time.perf_counter()
    while True:
        O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"],
                      region_dim=case["region_dim"], grid_hw=case["grid_hw"], vt_strides=[3.5, 7, 14, 28])
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20:
            break
    return dict(value=n / el, unit="images/s", cores=os.cpu_count(), kind="port",
                sample=f"{n} image(s) x {case['boxes'].shape[0]} boxes through oracle stages {Pipeline.stages} "
                       f"(oracle/hfre_oracle.py + roi_align_ref.c, OpenMP over boxes) in {el:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--boxes", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from vlm_fo1_amd import lib as L
    L.load()
    case = build_workload(dev, n_boxes=args.boxes, seed=1234 + rank)
    pipe = Pipeline(case)

    for _ in range(args.warmup):
        pipe.step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    # ---- roofline of the dominant kernel: separate profiled pass (hipEvents per launch) ----
    roof = None
    if rank == 0:
        L.profile(True)
        for _ in range(min(args.steps, 50)):
            pipe.step()
        torch.cuda.synchronize()
        rows = L.profile_rows(reset=True)
        L.profile(False)
        rows.sort(key=lambda r: -r["total_ms"])
        dom = rows[0]
        avg_ms = dom["total_ms"] / dom["calls"]
        work = dom["total_work"] / dom["calls"]
        ach = work / (avg_ms * 1e-3) / 1e9  # GB/s
        roof = dict(kernel=dom["name"], bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 5), traffic=None, avg_us=round(avg_ms * 1e3, 3),
                    algorithmic_bytes=work,
                    kernels={r["name"]: round(r["total_ms"] / r["calls"] * 1e3, 3) for r in rows})

    if rank == 0:
        n_img = args.steps * world
        out = dict(metric="images/sec", value=n_img / el, unit="images/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=el / args.steps * 1e3, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="bf16", data="synthetic",
                   region_tokens_per_sec=n_img * args.boxes / el,
                   config=dict(workload=f"BASELINE configs[1]: 1 image 640x480 x {args.boxes} proposals "
                                        "(CountBench UPN boxes), Qwen2.5-VL-3B / DaViT-L shapes, bf16 maps, fp32 region features",
                               stages=Pipeline.stages,
                               stages_not_yet_in_step=["qwen_vit", "davit", "simple_fpn", "mm_projector_aux", "llm_prefill"],
                               parallelism=f"dp{world} (images sharded, no data-path collective)"),
                   roofline=roof)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(case)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
