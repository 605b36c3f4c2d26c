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
    CountBench fixture item with N>=32, rescaled to the image), Qwen2.5-VL-3B / DaViT-L true shapes.
    Everything the timed region reads is resident in HBM."""
    from hfre_cases import box_fixtures
    from vlm_fo1_amd.model import synthetic_prompt
    H, W = img_hw
    g = torch.Generator().manual_seed(seed)
    gh, gw = round(H / 28) * 2, round(W / 28) * 2
    pix = torch.randn(gh * gw, 1176, generator=g).bfloat16()        # normalised patches, HF processor layout
    aux = torch.randn(3, H, W, generator=g).bfloat16()               # CLIP-normalised aux image ('dynamic': no resize)
    it = [x for x in box_fixtures()["countbench"] if len(x["bboxes"]) >= n_boxes][0]
    b = torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes]
    ex, ey = it["extent"]
    b = b * torch.tensor([W / ex, H / ey, W / ex, H / ey])
    ids = synthetic_prompt(n_boxes, n_text=60, seed=seed)
    case = dict(pix=pix, aux=aux, boxes=b, ids=ids, grid=(gh, gw), img_hw=img_hw)
    if device is not None:
        case["dev"] = dict(pix=pix.to(device), aux=aux.to(device), boxes=b.to(device))
    return case


class Pipeline:
    """Every stage of the hot path, back to back on one stream: one step = one image up to and including its first generated
    token.  `inflight` > 1: that many engine replicas (shared weights, private KV cache / graphs) on their own HIP streams, steps
    dealt round-robin, so independent images overlap on the GPU (a batch-1 pass under-fills 256 CUs)."""
    stages = ["qwen_vit(32 blocks)+merger", "mm_projector", "davit_large", "simple_fpn", "hfre_region_pool",
              "mm_projector_aux", "splice+mrope", "llm_prefill(36 layers)", "lm_head(last row)+argmax"]

    def __init__(self, case, device, inflight=1):
        from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
        self.cfg = FO1Config()
        self.weights = random_weights(self.cfg, device, seed=0)
        self.eng = FO1Engine(self.cfg, self.weights, device)
        self.engs = [self.eng] + [self.eng.replica() for _ in range(inflight - 1)]
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(inflight - 1)]
        self.case = case

    def step(self, graph=True, slot=0):
        d = self.case["dev"]
        with torch.cuda.stream(self.streams[slot]):
            return self.engs[slot].prefill(self.case["ids"], d["pix"], self.case["grid"], d["aux"], d["boxes"], use_graph=graph)


def cpu_baseline(case, pipe, budget_s=25.0):
    """The oracle (port) of the same stages on the host cores.  Bounded sample: the ViT and the LLM are timed on 9 / 8
    blocks and scaled by the block count (every block of a stage does identical work; the 8 extra ViT blocks hold one
    full-attention block, the model's 4-in-32 ratio); DaViT, SimpleFPN, HFRE and the projectors are timed in full."""
    import torch.nn.functional as F
    from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, vit_oracle as VO
    ncpu = min(32, len(os.sched_getaffinity(0)))  # 256-thread torch on this box thrashes; 32 is the fastest setting we measured
    torch.set_num_threads(ncpu)
    W = pipe.weights
    gh, gw = case["grid"]
    t = {}

    def cpu(sd, keep):
        return {k: v.float().cpu() for k, v in sd.items() if keep(k)}

    nv, nl = 9, 8
    vit_sd = cpu(W["vit"], lambda k: not k.startswith("blocks.") or int(k.split(".")[1]) < nv)
    t0 = time.perf_counter()
    tokens, maps = VO.vit_forward(vit_sd, case["pix"].float(), gh, gw, depth=nv, n_heads=16, fullatt=(1,))
    t["vit_blocks%d+embed+merger" % nv] = time.perf_counter() - t0
    t0 = time.perf_counter()
    VO.vit_forward(vit_sd, case["pix"].float(), gh, gw, depth=1, n_heads=16, fullatt=())
    t["vit_1block+embed+merger"] = time.perf_counter() - t0
    per_vit_block = max(t["vit_blocks%d+embed+merger" % nv] - t["vit_1block+embed+merger"], 1e-3) / (nv - 1)
    vit_total = t["vit_1block+embed+merger"] + per_vit_block * (pipe.cfg.vit.depth - 1)
    dav_sd = cpu(W["davit"], lambda k: True)
    for _ in range(2):   # second (warm) run is the one reported
        t0 = time.perf_counter()
        aux_maps, aux_sizes = DO.davit_forward(dav_sd, case["aux"].float().unsqueeze(0))
        t["davit"] = time.perf_counter() - t0
    fpn_sd = cpu(W["fpn"], lambda k: True)
    for _ in range(2):
        t0 = time.perf_counter()
        fpn = FO.fpn_forward(fpn_sd, maps[-1].reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
        t["fpn"] = time.perf_counter() - t0
    H, Wd = case["img_hw"]
    sw, sh = gw * 14 / Wd, gh * 14 / H
    aux_nchw = [m.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(aux_maps, aux_sizes)]
    t0 = time.perf_counter()
    feat = HO.hfre_oracle(aux_nchw, case["boxes"], fpn, case["boxes"] * torch.tensor([sw, sh, sw, sh]), region_dim=5888,
                          grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
    t["hfre"] = time.perf_counter() - t0
    proj = cpu(W["proj"], lambda k: True)
    t0 = time.perf_counter()
    reg = F.linear(F.gelu(F.linear(feat, proj["mm_projector_aux.0.weight"], proj["mm_projector_aux.0.bias"])),
                   proj["mm_projector_aux.2.weight"], proj["mm_projector_aux.2.bias"])
    img = F.linear(F.gelu(F.linear(tokens, proj["mm_projector.0.weight"], proj["mm_projector.0.bias"])),
                   proj["mm_projector.2.weight"], proj["mm_projector.2.bias"])
    t["projectors"] = time.perf_counter() - t0
    llm_sd = cpu(W["llm"], lambda k: not k.startswith("layers.") or int(k.split(".")[1]) < nl)
    emb, nb, na = LO.splice(torch.tensor(case["ids"]), llm_sd["embed_tokens.weight"], img, reg)
    pos, _ = LO.rope_index(nb, (gh // 2, gw // 2), na)
    t0 = time.perf_counter()
    fin = LO.llm_forward(llm_sd, emb, pos, n_layers=nl, n_heads=16, n_kv=2, head_dim=128, eps=1e-6, theta=1e6, sections=(16, 24, 24))
    t["llm_%dlayer" % nl] = time.perf_counter() - t0
    t0 = time.perf_counter()
    (fin[-1:] @ llm_sd["embed_tokens.weight"].t()).argmax()
    t["lm_head"] = time.perf_counter() - t0
    total = vit_total + t["davit"] + t["fpn"] + t["hfre"] + t["projectors"] + t["llm_%dlayer" % nl] / nl * pipe.cfg.llm.num_layers + t["lm_head"]
    return dict(value=1.0 / total, unit="images/s", cores=ncpu, host_cpus=os.cpu_count(), kind="port",
                seconds_per_image=round(total, 2), stage_seconds={k: round(v, 3) for k, v in t.items()},
                sample=f"1 image x {case['boxes'].shape[0]} boxes through the oracle stages on {ncpu} host threads (torch fp32): "
                       f"DaViT-L, SimpleFPN, HFRE, projectors, lm_head timed in full; ViT timed on {nv} of {pipe.cfg.vit.depth} blocks and "
                       f"the LLM on {nl} of {pipe.cfg.llm.num_layers} layers, scaled by block count")


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_traffic.json, written by scripts/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs with the
    gfx950 x2 FETCH correction of MI355X_MICROARCH.md applied).  PMC counters cannot be read from inside the process, so the
    bench line carries the committed figure and names its source; null when no PMC pass exists for the kernel."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        row = t["kernels"].get(kernel_name)
        if row:
            return row["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--boxes", type=int, default=32)
    ap.add_argument("--image", default="480x640", help="HxW of the synthetic image (default = BASELINE configs[1]; 1344x1344 with "
                    "--boxes 100 is the high-resolution configuration's geometry)")
    ap.add_argument("--inflight", type=int, default=3, help="independent single-image passes in flight per GPU (streams); 1 = strictly "
                    "one image at a time (latency mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the hipGraph")
    ap.add_argument("--profile-shapes", action="store_true", help="per-shape GEMM rows in roofline.per_step_ms")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # FO1_BENCH_ONE_DEVICE=1 (tests on a 1-GPU box only): every rank on cuda:0, rendezvous over gloo — exercises the multi-rank
    # control flow (barriers, max-over-ranks, rank-0 line); RCCL refuses two ranks on one device.
    one_dev = os.environ.get("FO1_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from vlm_fo1_amd import lib as L
    L.load()
    img_hw = tuple(int(v) for v in args.image.lower().split("x"))
    case = build_workload(dev, n_boxes=args.boxes, img_hw=img_hw, seed=1234 + rank)
    R = max(1, args.inflight)
    pipe = Pipeline(case, dev, inflight=R)

    use_graph = not args.eager
    for slot in range(R):
        for _ in range(args.warmup):
            pipe.step(use_graph, slot)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.step(use_graph, k % R)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cpu" if one_dev else dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    # ---- latency mode: strictly one image at a time on one stream (not `value` unless --inflight 1) ----
    single = None
    if rank == 0:
        if R == 1:
            single = dict(images_per_sec=args.steps / el, ms_per_image=el / args.steps * 1e3)
        else:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step(use_graph, 0)
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            single = dict(images_per_sec=round(args.steps / e1, 2), ms_per_image=round(e1 / args.steps * 1e3, 3))

    # ---- greedy decode through the KV cache (fixed K new tokens, reported separately; not part of `value`) ----
    dec = None
    if rank == 0:
        K = 32
        out = pipe.step(use_graph)
        tok = out["next_token"]
        llm = pipe.eng.llm
        if use_graph:
            llm.sync_decode_state()
            _, tok = llm.decode_step_graph(tok)
            for _ in range(2):
                llm.decode_step_graph()
            torch.cuda.synchronize()
            td = time.perf_counter()
            for _ in range(K):
                _, tok = llm.decode_step_graph()
                tok.item()                      # the host reads every token (stop criteria), as generate() does
            td = (time.perf_counter() - td) / K
        else:
            for _ in range(3):
                _, _, tok = llm.decode_step(tok)
            torch.cuda.synchronize()
            td = time.perf_counter()
            for _ in range(K):
                _, _, tok = llm.decode_step(tok)
                tok.item()
            td = (time.perf_counter() - td) / K
        dec = dict(ms_per_token=round(td * 1e3, 3), tokens_per_sec=round(1.0 / td, 1), new_tokens_timed=K,
                   weight_stream_floor_ms=round(6.2e9 / 8e12 * 1e3, 3),
                   images_per_sec_with_64_token_answer=round(1.0 / (single["ms_per_image"] * 1e-3 + 64 * td), 2))

    # ---- host-side preprocessing of one image (SURVEY 8d "preprocess (CPU)" stage, 8f rank 2): not part of `value` ----
    prep = None
    if rank == 0:
        import numpy as np
        from PIL import Image
        from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor, Qwen2VLPatchProcessor
        H, W = case["img_hw"]
        pil = Image.fromarray(np.random.default_rng(1234).integers(0, 256, (H, W, 3), dtype=np.uint8), "RGB")
        prep = {}
        for label, on_dev in (("host_fp32_then_upload", False), ("uint8_upload_then_device_kernels", True)):
            p1, p2 = Qwen2VLPatchProcessor(), CLIPStyleAuxProcessor(resize_mode="dynamic")
            if on_dev:
                p1.device = p2.device = dev
            for it in range(12):
                if it == 2:
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                a = p1.preprocess(pil, return_tensors="pt")["pixel_values"].to(dev, dtype=torch.bfloat16)
                b = p2.preprocess(pil, return_tensors="pt")["pixel_values"][0].to(dev, dtype=torch.bfloat16)
            torch.cuda.synchronize()
            prep[label + "_ms"] = round((time.perf_counter() - tp) / 10 * 1e3, 3)
        prep["note"] = "PIL image (already decoded, no resize needed at 640x480) -> both towers' bf16 device tensors"

    # ---- roofline of the dominant kernel: separate profiled pass (hipEvents per launch) ----
    roof = None
    if rank == 0:
        L.load().fo1_gemm_profile_shapes(1 if args.profile_shapes else 0)
        nprof = min(args.steps, 50)
        TAGS = {"qwen_vit+merger": "vit", "mm_projector": "proj", "davit_large": "davit", "simple_fpn": "fpn",
                "hfre_region_pool": "hfre", "mm_projector_aux": "proj_aux", "splice": "splice", "llm_prefill+lm_head+argmax": "llm"}
        L.profile(True)
        pipe.eng.stage_hook = lambda stage: L.profile_stage(TAGS[stage])   # tags records, no sync
        per_step, per_stage = [], []
        inv = {v: k for k, v in TAGS.items()}
        for _ in range(nprof):
            pipe.step(graph=False)   # per-kernel timestamps need individual launches, not a graph replay
            torch.cuda.synchronize()
            # drain once per step (<= ~1000 event pairs outstanding); the per-kernel figure is the MEDIAN over steps of the
            # step's total for that kernel, so a sporadic stall in one step does not leak into the average
            ks, st = {}, {}
            for r in L.profile_rows(reset=True):
                tag, _, kname = r["name"].rpartition("|")
                stage = inv.get(tag, "unattributed")
                st[stage] = st.get(stage, 0.0) + r["total_ms"]
                m = ks.setdefault(kname, dict(name=kname, calls=0, total_ms=0.0, total_work=0.0))
                m["calls"] += r["calls"]
                m["total_ms"] += r["total_ms"]
                m["total_work"] += r["total_work"]
            per_step.append(ks)
            per_stage.append(st)
        pipe.eng.stage_hook = None
        L.profile(False)

        def median(v):
            v = sorted(v)
            return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

        rows = []
        for kname, first in per_step[0].items():   # every step launches the same kernels
            rows.append(dict(name=kname, calls=first["calls"], total_work=first["total_work"],
                             total_ms=median([ks[kname]["total_ms"] for ks in per_step if kname in ks])))
        rows.sort(key=lambda r: -r["total_ms"])
        dom = rows[0]
        stage_ms = {k: round(median([st.get(k, 0.0) for st in per_stage]), 4) for k in per_stage[0]}
        nprof = 1   # rows now hold ONE step's launches / work with the median step time
        avg_ms = dom["total_ms"] / dom["calls"]
        work = dom["total_work"] / dom["calls"]
        mfma = dom["name"].startswith("gemm") or dom["name"].startswith("attn")
        ach = work / (avg_ms * 1e-3) / (1e12 if mfma else 1e9)
        peak = MFMA_BF16_PEAK_TF if mfma else HBM_PEAK_GBS
        traffic, traffic_src = pmc_traffic(dom["name"])
        # all MFMA GEMM templates together (the three tile shapes are one kernel source)
        g_rows = [r for r in rows if r["name"].startswith("gemm_bt_")]
        g_ms = sum(r["total_ms"] for r in g_rows)
        g_tf = sum(r["total_work"] for r in g_rows) / (g_ms * 1e-3) / 1e12 if g_ms > 0 else None
        roof = dict(kernel=dom["name"], bound="mfma" if mfma else "hbm", achieved=round(ach, 2), peak=peak,
                    unit="TFLOP/s" if mfma else "GB/s", frac=round(ach / peak, 5), traffic=traffic,
                    traffic_source=traffic_src,
                    all_gemm_tiles=dict(tflops=round(g_tf, 2) if g_tf else None, ms_per_step=round(g_ms / nprof, 3),
                                        launches_per_step=sum(r["calls"] for r in g_rows) // nprof),
                    avg_us=round(avg_ms * 1e3, 3), launches_per_step=dom["calls"] // nprof,
                    algorithmic_work_per_launch=work,
                    per_step_ms={r["name"]: round(r["total_ms"] / nprof, 4) for r in rows},
                    launches={r["name"]: r["calls"] // nprof for r in rows})
        # the HFRE gather is the HBM-bound kernel north_star names: report it next to the dominant (MFMA) kernel
        by = {r["name"]: r for r in rows}
        if "hfre_pool" in by:
            pool = by["hfre_pool"]
            t_all = sum(by[k]["total_ms"] for k in ("hfre_weights", "hfre_pool", "hfre_finish") if k in by)
            hb, hsrc = pmc_traffic("hfre_pool")
            roof["hfre"] = dict(bound="hbm", kernel="hfre_pool (+ hfre_weights, hfre_finish)", peak=HBM_PEAK_GBS, unit="GB/s",
                                algorithmic_bytes=pool["total_work"] / pool["calls"],
                                achieved_pool_kernel=round(pool["total_work"] / pool["calls"] / (pool["total_ms"] / pool["calls"] * 1e-3) / 1e9, 1),
                                achieved_all_three=round(pool["total_work"] / pool["calls"] / (t_all / pool["calls"] * 1e-3) / 1e9, 1),
                                frac=round(pool["total_work"] / pool["calls"] / (pool["total_ms"] / pool["calls"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                us_pool=round(pool["total_ms"] / pool["calls"] * 1e3, 2), us_all_three=round(t_all / pool["calls"] * 1e3, 2),
                                traffic=hb, traffic_source=hsrc,
                                note="algorithmic bytes = every source map once + output (SURVEY 8d upper bound); the boxes' footprints "
                                     "cover about a third of it, so the kernel is latency-bound at this size, not bandwidth-bound")

    if rank == 0:
        n_img = args.steps * world
        out = dict(metric="images/sec", value=n_img / el, unit="images/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=el / args.steps * 1e3, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="bf16", data="synthetic",
                   region_tokens_per_sec=n_img * args.boxes / el,
                   config=dict(workload=f"{'BASELINE configs[1]' if (img_hw == (480, 640) and args.boxes == 32) else 'non-default geometry'}: 1 image "
                                        f"{img_hw[1]}x{img_hw[0]} (S={case['grid'][0] * case['grid'][1]} patches) x {args.boxes} proposals "
                                        f"(CountBench UPN boxes), Qwen2.5-VL-3B + DaViT-L + SimpleFPN true shapes, prompt "
                                        f"{len(case['ids']) - 1 + case['grid'][0] * case['grid'][1] // 4} tokens after splice, prefill to the first greedy token",
                               stages=Pipeline.stages,
                               launch=("eager" if args.eager else "hipGraph replay (1 graph per shape signature)") +
                                      (f"; {R} independent images in flight on {R} HIP streams (engine replicas share weights)" if R > 1 else "; one image at a time"),
                               images_in_flight=R,
                               parallelism=f"dp{world} (images sharded, no data-path collective)" + (" [test: all ranks on one device]" if one_dev else "")),
                   one_image_at_a_time=single, decode=dec, preprocess=prep, roofline=roof)
        if roof is not None:
            # stage times (sum of kernel execution time per stage, eager pass) and the two region-token rates
            out["stage_kernel_ms"] = stage_ms
            t_reg = stage_ms.get("hfre_region_pool", 0.0) + stage_ms.get("mm_projector_aux", 0.0)
            t_enc = t_reg + stage_ms.get("davit_large", 0.0) + stage_ms.get("simple_fpn", 0.0)
            if t_reg > 0:
                out["region_tokens_per_sec_hfre_plus_connector"] = round(args.boxes / (t_reg * 1e-3), 1)
                out["region_tokens_per_sec_encode_regions"] = round(args.boxes / (t_enc * 1e-3), 1)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(case, pipe)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()   # ranks leave together (rank 0 was still measuring decode / roofline)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
