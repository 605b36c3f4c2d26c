#!/usr/bin/env python3
"""bench.py — throughput of the VLM-FO1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = ONE packed pass of `--batch` (default 25) different images through every hot-path stage the engine implements
(listed in config.stages: both towers, FPN, HFRE, connectors, splice, 36-layer LLM prefill, first greedy token), inputs already
resident in HBM; `--inflight` (default 2) passes are in flight on their own HIP streams.  value = images / second.  N>1: images
shard across ranks with no data-path collective (weak scaling); value = images all ranks processed / max-over-ranks time.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, live dispatch timestamps), `cpu_baseline` (the oracle at full depth
on this box's host cores, N = 1 only) and side measurements (one pass / one image at a time, decode loops, preprocessing;
`--main-only` skips them).  `--fp8` is a secondary, opt-in measurement (e4m3 linears); the default line is bf16.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16
MFMA_FP8_PEAK_TF = 5000.0   # dense fp8 (the secondary --fp8 measurement's dominant kernel)


def build_workload(device, n_boxes=100, img_hw=(480, 640), seed=1234, lift_cap=False):
    """The configuration BASELINE.json's `metric` is quoted on: 1 image (640x480 synthetic, the COCO-typical size) x 100
    proposals (the reference's cap, mm_utils.py:600; boxes = the 100-box CountBench UPN fixture item rescaled to the image),
    Qwen2.5-VL-3B / DaViT-L true shapes.  n_boxes=32 gives configs[1].  Everything the timed region reads is resident in HBM."""
    from vlm_fo1_amd.fixtures import box_fixtures
    from vlm_fo1_amd.model import synthetic_prompt
    H, W = img_hw
    g = torch.Generator().manual_seed(seed)
    gh, gw = round(H / 28) * 2, round(W / 28) * 2
    pix = torch.randn(gh * gw, 1176, generator=g).bfloat16()        # normalised patches, HF processor layout
    aux = torch.randn(3, H, W, generator=g).bfloat16()               # CLIP-normalised aux image ('dynamic': no resize)
    fx = box_fixtures()
    if n_boxes <= 100:
        it = [x for x in fx["countbench"] if len(x["bboxes"]) >= n_boxes][0]
        b = torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes]
        ex, ey = it["extent"]
        b = b * torch.tensor([W / ex, H / ey, W / ex, H / ey])
    else:
        # more proposals than the reference's cap of 100 features per prompt (mm_utils.py:600; a longer prompt IndexErrors at
        # omchat_qwen2_5_vl.py:361): the boxes of several fixture items, run as ceil(N / 100) prompts of <= 100 over the SAME image
        chunks = []
        for it in sorted(fx["countbench"] + fx["pixmo"], key=lambda x: -len(x["bboxes"])):
            ex, ey = it["extent"]
            chunks.append(torch.tensor(it["bboxes"], dtype=torch.float32)[:100] * torch.tensor([W / ex, H / ey, W / ex, H / ey]))
        b = torch.cat(chunks)[:n_boxes]
        assert b.shape[0] == n_boxes, f"the fixtures hold {torch.cat(chunks).shape[0]} boxes, {n_boxes} asked"
    ids = synthetic_prompt(n_boxes if lift_cap else min(n_boxes, 100), n_text=60, seed=seed)
    case = dict(pix=pix, aux=aux, boxes=b, ids=ids, grid=(gh, gw), img_hw=img_hw)
    if device is not None:
        case["dev"] = dict(pix=pix.to(device), aux=aux.to(device), boxes=b.to(device))
    if n_boxes > 100 and not lift_cap:
        case["prompts"] = [(synthetic_prompt(min(100, n_boxes - k), n_text=60, seed=seed + k, lead_seed=seed), b[k:k + 100]) for k in range(0, n_boxes, 100)]     # one preamble per image, another question per prompt
    return case


class Pipeline:
    """Every stage of the hot path, back to back on one stream: one step = one image up to and including its first generated
    token.  `inflight` > 1: that many engine replicas (shared weights, private KV cache / graphs) on their own HIP streams, steps
    dealt round-robin, so independent images overlap on the GPU (a batch-1 pass under-fills 256 CUs)."""
    stages = ["qwen_vit(32 blocks)+merger", "mm_projector", "davit_large", "simple_fpn", "hfre_region_pool",
              "mm_projector_aux", "splice+mrope", "llm_prefill(36 layers)", "lm_head(last row)+argmax"]

    def __init__(self, case, device, inflight=1, batch=1, cases=None):
        from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
        self.cfg = FO1Config()
        self.weights = random_weights(self.cfg, device, seed=0)
        self.eng = FO1Engine(self.cfg, self.weights, device)
        self.engs = [self.eng] + [self.eng.replica() for _ in range(inflight - 1)]
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(inflight - 1)]
        self.case = case
        self.batch = batch
        # `batch` DIFFERENT images of the same geometry per step (cases[i]); packed into one pass of every stage
        self.cases = cases if cases is not None else [case] * batch
        self.requests = []
        for ci, c in enumerate(self.cases):
            if "prompts" in c:      # several prompts over one image (> 100 proposals): the towers run once per image (image_id)
                for ids, bx in c["prompts"]:
                    self.requests.append(dict(ids=ids, pix=c["dev"]["pix"], grid=c["grid"], aux=c["dev"]["aux"], boxes=bx.to(device), image_id=ci))
            else:
                self.requests.append(dict(ids=c["ids"], pix=c["dev"]["pix"], grid=c["grid"], aux=c["dev"]["aux"], boxes=c["dev"]["boxes"]))
        self.multi_prompt = any("prompts" in c for c in self.cases)

    def step(self, graph=True, slot=0):
        """One step = one packed pass over `batch` images (batch 1: one image)."""
        with torch.cuda.stream(self.streams[slot]):
            if self.batch == 1 and not self.multi_prompt:
                d = self.case["dev"]
                return self.engs[slot].prefill(self.case["ids"], d["pix"], self.case["grid"], d["aux"], d["boxes"], use_graph=graph)
            return self.engs[slot].prefill_batch(self.requests, use_graph=graph)

    def step_single(self, graph=True):
        """Latency mode: ONE image through the same stages (a batch of one)."""
        if self.multi_prompt:
            n = len(self.cases[0]["prompts"])
            return self.eng.prefill_batch(self.requests[:n], use_graph=graph)
        d = self.case["dev"]
        return self.eng.prefill(self.case["ids"], d["pix"], self.case["grid"], d["aux"], d["boxes"], use_graph=graph)


def dataset_run(pipe, name, n_items, batch, inflight, aux_mode="dynamic", decode_tokens=0, row_budget=None):
    """Dataset-shaped side measurement (BASELINE configs[2] / configs[3]): the items of `name`
    (bench_workloads.py: the reference's CountBench / Pixmo fixtures verbatim, or COCO-like sizes x 100 boxes) with their OWN image
    sizes and box counts, packed by cost into passes of <= `batch` images / `row_budget` ViT rows, `inflight` passes in flight.
    Every pass has a new shape signature, so passes launch eagerly (no graph replay).  One untimed sweep (scratch allocation, index
    plans), then one timed sweep.  `uniform_equivalent_images_per_sec` prices the same work in metric-configuration images
    (1564 patches each) so it can be read against `value`."""
    import bench_workloads as BW
    dev = pipe.eng.dev
    reqs, geos = BW.build_requests(name, dev, limit=n_items, aux_mode=aux_mode)
    # packing policy (profiles/r03_ragged_profile_countbench.json): up to 64 images and ~40k ViT rows per pass — the uniform pass's
    # row count; small images then still give the GEMMs a full M (25 thumbnails per pass left 3/4 of the tiles empty)
    groups = BW.pack(geos, batch=max(batch, 64), row_budget=row_budget or 40000)
    need = max(sum(len(reqs[i]["ids"]) + geos[i]["S"] // 4 + 8 for i in g) for g in groups)
    for e in pipe.engs:
        if e.llm.reserve(need):
            e._graphs.clear()
            e._seen.clear()

    def sweep():
        for k, g in enumerate(groups):
            slot = k % inflight
            with torch.cuda.stream(pipe.streams[slot]):
                grp = [reqs[i] for i in g]
                if decode_tokens:
                    pipe.engs[slot].generate_batch(grp, max_new_tokens=decode_tokens, use_graph=True)
                else:
                    pipe.engs[slot].prefill_batch(grp, use_graph=False)   # a dataset's signatures do not repeat: nothing to replay
        torch.cuda.synchronize()

    sweep()
    t0 = time.perf_counter()
    sweep()
    el = time.perf_counter() - t0
    patches = sum(g["S"] for g in geos)
    out = dict(BW.summary(geos), name=name, aux=aux_mode, passes=len(groups), images_per_pass_max=max(batch, 64), passes_in_flight=inflight,
               seconds=round(el, 3), images_per_sec=round(len(reqs) / el, 2), region_tokens_per_sec=round(sum(g["n"] for g in geos) / el, 1),
               uniform_equivalent_images_per_sec=round(patches / 1564.0 / el, 2), launch="eager (every pass is a new shape signature)",
               packing="<= 64 images and <= 40k ViT rows per pass, items sorted by cost",
               what="prefill to the first greedy token" if not decode_tokens else f"prefill + {decode_tokens}-token batched greedy decode")
    return out


def end_to_end_run(pipe, cases, steps, K=64, pool_slots=0, pools=None):
    """END-TO-END images/s ("images completed / wall time", decode K reported) in ONE timed loop per pass of len(cases)
    images: resized uint8 images in pinned host memory -> upload -> device preprocessing of both towers (patchify / normalise,
    fo1_patchify_u8_bf16 / fo1_normalize_u8_bf16) -> ONE packed prefill pass -> K greedy tokens per image in the batched device decode
    loop (groups of BatchDecoder.MAX_BATCH sequences, stop rule on the device; random weights never emit a stop id, so every image
    decodes exactly K tokens) -> generated ids on the host.  One host thread per engine replica / HIP stream (generate_batch blocks
    on the ids), `len(pipe.engs)` passes in flight.  Outside: JPEG decode and the bicubic resize (PIL, host: the job of
    sharded_eval.Prefetcher's threads) and tokenisation (cached prefix).
    pool_slots = 64 / 128: CONTINUOUS BATCHING (vlm_fo1_amd/serving.py) — the passes' sequences join ONE decode pool of that many
    slots as soon as their prefill is done and the replica goes straight on with its next pass; sequences of up to pool_slots / 32
    passes share every decode step (one weight stream per step for all of them).  0: every pass decodes its own group (round 3)."""
    import threading
    import numpy as np
    from vlm_fo1.model.image_processing import IMAGENET_MEAN, IMAGENET_STD, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, normalise_lut
    from vlm_fo1_amd import ops
    dev = pipe.eng.dev
    rng = np.random.default_rng(7)
    hosts = []
    for c in cases:
        H, W = c["img_hw"]
        gh, gw = c["grid"]
        prim = torch.from_numpy(rng.integers(0, 256, (gh * 14, gw * 14, 3), dtype=np.uint8)).pin_memory()     # after smart-resize
        aux = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).pin_memory()                   # 'dynamic': the image itself
        hosts.append((prim, aux))
    lut_p = normalise_lut(OPENAI_CLIP_MEAN, OPENAI_CLIP_STD).to(dev)
    lut_a = normalise_lut(IMAGENET_MEAN, IMAGENET_STD).to(dev)
    n_tok = [0]

    svc = None
    if pool_slots:
        svc = pipe.eng.enable_decode_pool(slots=pool_slots, pools=pools)
        for e in pipe.engs:
            e._pool_svc = svc

    def requests_of_pass():
        reqs = []
        for c, (prim, aux) in zip(cases, hosts):
            pu, au = prim.to(dev, non_blocking=True), aux.to(dev, non_blocking=True)
            reqs.append(dict(ids=c["ids"], pix=ops.patchify_u8(pu, lut_p, 14, 2), grid=c["grid"], aux=ops.normalize_u8(au, lut_a), boxes=c["dev"]["boxes"]))
        return reqs

    def one_pass(slot, handles=None):
        eng = pipe.engs[slot]
        with torch.cuda.stream(pipe.streams[slot]):
            reqs = requests_of_pass()
            if handles is not None:          # pool: hand the sequences over, go on with the next pass; ids are collected at the end
                handles.append(eng.submit_batch(reqs, max_new_tokens=K, use_graph=True))
                return 0
            ids = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)
            assert all(len(t) == K for t in ids)
            return sum(len(t) for t in ids)

    R = len(pipe.engs)
    for _ in range(2):              # untimed, twice: the second sighting of a pass shape captures its hipGraph (and the decode graphs)
        for slot in range(R):
            one_pass(slot)
    torch.cuda.synchronize()
    per = max(1, steps // R)

    lock = threading.Lock()

    def worker(slot):
        torch.cuda.set_device(dev)
        hs = [] if svc is not None else None
        n = 0
        for _ in range(per):
            n += one_pass(slot, hs)
        for h in hs or ():
            ids = h.result(timeout=600)
            assert all(len(t) == K for t in ids)
            n += sum(len(t) for t in ids)
        with lock:
            n_tok[0] += n

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(s,)) for s in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_img = per * R * len(cases)
    from vlm_fo1_amd.llm import BatchDecoder
    extra = {}
    if svc is not None:
        st = dict(svc.stats)
        extra = dict(decode="continuous batching: one decode pool per GPU, sequences of successive prefill passes share every step",
                     pool_slots=pool_slots, pools=st.get("pools", 1), pool_steps=st["steps"], pool_mean_live_sequences=round(st["occupancy_sum"] / max(1, st["steps"]), 1))
        pipe.eng.disable_decode_pool()
        for e in pipe.engs:
            e._pool_svc = None
    return dict(images_per_sec=round(n_img / el, 2), new_tokens_per_image=K, generated_tokens_per_sec=round(n_tok[0] / el, 1),
                ms_per_pass=round(el / (per * R) * 1e3 * R, 3), passes_timed=per * R, images_per_pass=len(cases), passes_in_flight=R,
                decode_group=pool_slots or BatchDecoder.MAX_BATCH, **extra,
                includes=["uint8 upload (pinned)", "device preprocessing (both towers)", "packed prefill", f"{K}-token batched greedy decode", "ids to host"],
                excludes=["JPEG decode + bicubic resize (host prefetch threads)", "tokenisation (cached prefix)"])


def driver_level_run(pipe, n_items=768, n_warm=128, K=64, batch=32, inflight=2, pool_slots=128, prefetch_threads=4):
    """DRIVER-LEVEL images/s: THIS REPO's `evaluation/eval_coco.py` eval_coco() as a user runs it — the reference's
    evaluation loop (reference evaluation/eval_coco.py:36-66: file -> PIL -> prepare_inputs -> generate -> decode -> regex -> COCO records
    -> json dump: same CLI, inputs and output file) RESTRUCTURED around the engine: prefetch threads, packed passes through
    generate_many_async, the decode pool, run_sharded.  (The reference's literal one-image-at-a-time loop on this engine is the line's
    `decode.images_per_sec_with_64_token_answer`, repeated in this block as `reference_literal_batch1_loop_images_per_sec`.)  Run on
    `n_items` synthetic 640 x 480 JPEG files x 100 UPN boxes, a deterministic stand-in tokenizer and the engine whose passes this bench
    has just timed (no checkpoint / tokenizer / dataset exists offline; `load_pretrained_model` is the
    one thing replaced).  Everything a user waits for is inside the timed call: jsonl parsing, image-header cost model, JPEG decode +
    bicubic resize + tokenisation + uploads on the prefetch threads, packed prefill passes on `inflight` worker threads, the decode
    pool, tokenizer.decode, the regex extraction and the dump.  max_new_tokens is fixed at K ($FO1_MAX_NEW_TOKENS; the reference's
    4096 relies on an EOS random weights never emit).  One untimed call first (graph captures, replicas, pool)."""
    import shutil
    import tempfile
    from vlm_fo1.model.fo1_model import FO1ForCausalLM, FO1HFConfig
    from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor, Qwen2VLPatchProcessor
    from vlm_fo1_amd.fixtures.synthetic import ToyTokenizer, full_config_dict, write_coco_like_dataset
    sys.path.insert(0, os.path.join(ROOT, "evaluation"))
    import eval_coco as E
    dev = pipe.eng.dev
    root = tempfile.mkdtemp(prefix="fo1_driver_level_")
    try:
        warm = write_coco_like_dataset(os.path.join(root, "warm"), n_warm, seed=1)
        data = write_coco_like_dataset(os.path.join(root, "data"), n_items, seed=2)
        model = FO1ForCausalLM.from_engine(FO1HFConfig(full_config_dict()), pipe.eng)
        primary, aux = Qwen2VLPatchProcessor(min_pixels=56 * 56, max_pixels=2048 * 2048), CLIPStyleAuxProcessor(size=768, resize_mode="dynamic")
        primary.device = aux.device = model.device
        tok = ToyTokenizer()
        E.load_pretrained_model = lambda model_id, device="cuda": (tok, model, (primary, aux))
        env = dict(FO1_BATCH=str(batch), FO1_INFLIGHT=str(inflight), FO1_DECODE_POOL=str(pool_slots), FO1_MAX_NEW_TOKENS=str(K),
                   FO1_PREFETCH_THREADS=str(prefetch_threads))
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        name = "synthetic/VLM-FO1_Qwen2.5-VL-3B-synthetic"
        import contextlib
        try:
            # the drivers (like the reference's) print every prompt: this process's stdout carries ONE JSON line, so theirs goes to a file
            with open(os.path.join(root, "driver_stdout.log"), "w") as log, contextlib.redirect_stdout(log):
                E.eval_coco(name, warm[0], warm[1], warm[2], os.path.join(root, "out_warm"), device=str(dev))
                torch.cuda.synchronize()
                import gc
                gc.collect()
                gc.freeze()      # what load_pretrained_model does for a served engine (vlm_fo1/model/builder.py): no gen-2 sweeps over its object graph
                t0 = time.perf_counter()
                E.eval_coco(name, data[0], data[1], data[2], os.path.join(root, "out"), device=str(dev))
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            # host side of one item, single thread (what the prefetch threads do): sizes the threads an 8-GPU node needs
            from vlm_fo1.mm_utils import prepare_inputs
            import json as _json
            items = [_json.loads(l) for l in open(data[0])][:24]
            th = time.perf_counter()
            with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
                for d in items:
                    messages = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": os.path.join(data[2], d["image"])}},
                                                             {"type": "text", "text": d["conversations"][0]["value"]}], "bbox_list": d["bbox_list"]}]
                    prepare_inputs(name, model, (primary, aux), tok, messages, device=str(dev), max_tokens=K, top_p=0.05, temperature=0.0, do_sample=False)
            torch.cuda.synchronize()
            host_ms = (time.perf_counter() - th) / len(items) * 1e3
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            pipe.eng.disable_decode_pool()
            for r, _ in model.__dict__.get("_worker_replicas", []):
                r.engine._pool_svc = None
        out_file = os.path.join(root, "out", name.split("/")[-1], "eval_predictions.json")
        ips = n_items / el
        return dict(images_per_sec=round(ips, 2), items=n_items, seconds=round(el, 3), new_tokens_per_image=K, images_per_pass=batch,
                    prefill_workers=inflight, decode_pool_slots=pool_slots, prefetch_threads=prefetch_threads,
                    host_threads_per_gpu=inflight + prefetch_threads + 1 + (1 if pool_slots else 0),
                    host_prepare_ms_per_item_one_thread=round(host_ms, 2),
                    host_prepare_threads_needed_at_this_rate=round(ips * host_ms / 1e3, 2),
                    host_prepare_threads_needed_for_8_gpus=round(8 * ips * host_ms / 1e3, 1),
                    predictions_file_written=os.path.exists(out_file),
                    loop="evaluation/eval_coco.py eval_coco(): jsonl -> cost model -> [prefetch threads: PIL decode + resize + tokenise + upload + device "
                         "preprocess] -> packed prefill passes -> decode pool -> tokenizer.decode -> regex -> COCO records -> json dump",
                    data="synthetic 640x480 JPEG files x 100 UPN boxes; ToyTokenizer; random weights at the true shapes")
    finally:
        shutil.rmtree(root, ignore_errors=True)


def _write_count_dataset(root, name, limit=None, threads=16):
    """The CountBench / Pixmo-Count fixture as FILES in the format evaluation/eval_countbench.py reads (reference eval_countbench.py:14-30: a
    json list of {image, question, bboxes, answer}): every item of bench_workloads.dataset_items(name) — the reference's UPN box lists
    verbatim — with a JPEG synthesised at the extent of its boxes (smooth noise: decodes at a photo's cost).  -> (json path, image folder)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    import bench_workloads as BW
    items = BW.dataset_items(name, limit)
    img_dir = os.path.join(root, "images")
    os.makedirs(img_dir, exist_ok=True)

    def one(i):
        it = items[i]
        w, h = max(28, it["width"]), max(28, it["height"])
        small = np.random.default_rng(1000 + i).integers(0, 256, (max(4, h // 8), max(4, w // 8), 3), dtype=np.uint8)
        Image.fromarray(small, "RGB").resize((w, h), Image.BICUBIC).save(os.path.join(img_dir, f"{i:05d}.jpg"), quality=90)

    with ThreadPoolExecutor(max_workers=threads) as ex:      # PIL releases the GIL in resize / encode
        list(ex.map(one, range(len(items))))
    recs = [dict(image=f"{i:05d}.jpg", question="How many objects are there in the image?", bboxes=[[float(v) for v in b] for b in it["boxes"]],
                 answer=int(len(it["boxes"]))) for i, it in enumerate(items)]
    path = os.path.join(root, f"{name}.json")
    json.dump(recs, open(path, "w"))
    return path, img_dir, items


def driver_level_count_run(pipe, names=("countbench", "pixmo"), limit=None, K=64, batch=32, inflight=2, pool_slots=128, prefetch_threads=4):
    """DRIVER-LEVEL twin for BASELINE configs[3] (VERDICT r4 missing #5): `evaluation/eval_countbench.py`'s own loop (reference
    evaluation/eval_countbench.py:14-65: json -> per item PIL -> prepare_inputs -> generate -> decode -> first integer -> accuracy) on the
    FULL CountBench (487 items) and Pixmo-Count (529 items) fixtures as JPEG files of their own sizes (99 x 99 ... 5181 x 3444) and box counts
    (2 ... 100), through the sharded path (cost model, prefetch threads, ragged packed passes, decode pool).  Accuracy is meaningless on
    random weights; what is timed is everything a user waits for."""
    import contextlib
    import shutil
    import tempfile
    from vlm_fo1.model.fo1_model import FO1ForCausalLM, FO1HFConfig
    from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor, Qwen2VLPatchProcessor
    from vlm_fo1_amd.fixtures.synthetic import ToyTokenizer, full_config_dict
    sys.path.insert(0, os.path.join(ROOT, "evaluation"))
    import eval_countbench as E
    dev = pipe.eng.dev
    root = tempfile.mkdtemp(prefix="fo1_driver_count_")
    out = {}
    try:
        model = FO1ForCausalLM.from_engine(FO1HFConfig(full_config_dict()), pipe.eng)
        primary, aux = Qwen2VLPatchProcessor(min_pixels=56 * 56, max_pixels=2048 * 2048), CLIPStyleAuxProcessor(size=768, resize_mode="dynamic")
        primary.device = aux.device = model.device
        tok = ToyTokenizer()
        E.load_pretrained_model = lambda model_id, device="cuda": (tok, model, (primary, aux))
        env = dict(FO1_BATCH=str(batch), FO1_INFLIGHT=str(inflight), FO1_DECODE_POOL=str(pool_slots), FO1_MAX_NEW_TOKENS=str(K),
                   FO1_PREFETCH_THREADS=str(prefetch_threads))
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        name = "synthetic/VLM-FO1_Qwen2.5-VL-3B-synthetic"
        try:
            with open(os.path.join(root, "driver_stdout.log"), "w") as log, contextlib.redirect_stdout(log):
                for ds in names:
                    tw = time.perf_counter()
                    path, img_dir, items = _write_count_dataset(os.path.join(root, ds), ds, limit)
                    t_files = time.perf_counter() - tw
                    if not out:      # one untimed call on a small prefix first: replicas, the pool, the kernels' first launches
                        warm = os.path.join(root, ds, "warm.json")
                        json.dump(json.load(open(path))[:48], open(warm, "w"))
                        E.eval_countbench(warm, img_dir, name, str(dev))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    acc = E.eval_countbench(path, img_dir, name, str(dev))
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t0
                    npx = sum(it["width"] * it["height"] for it in items)
                    out[ds] = dict(images_per_sec=round(len(items) / el, 2), items=len(items), seconds=round(el, 3),
                                   boxes_per_item_mean=round(sum(len(it["boxes"]) for it in items) / len(items), 1),
                                   megapixels_per_item_mean=round(npx / len(items) / 1e6, 2), accuracy_on_random_weights=acc,
                                   seconds_writing_the_files_untimed=round(t_files, 1))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            pipe.eng.disable_decode_pool()
            for r, _ in model.__dict__.get("_worker_replicas", []):
                r.engine._pool_svc = None
        out["new_tokens_per_image"] = K
        out["loop"] = ("evaluation/eval_countbench.py eval_countbench(): json -> cost model -> [prefetch threads: JPEG decode + resize + tokenise + upload + device "
                       "preprocess] -> ragged packed prefill passes -> decode pool -> tokenizer.decode -> first integer -> accuracy")
        out["data"] = "the reference's CountBench / Pixmo-Count UPN box fixtures verbatim; JPEG files synthesised at the extent of each item's boxes; ToyTokenizer; random weights"
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def scale_run(pipe, world, rank, one_dev, items_per_gpu=256, n_warm_per_gpu=64, K=64, batch=32, inflight=2, pool_slots=128, prefetch_threads=4, sample=64):
    """`scale` block of a multi-rank run: the path north_star describes, timed ACROSS the ranks — evaluation/eval_coco.py's
    own loop (reference evaluation/eval_coco.py:36-88) through sharded_eval.run_sharded: LPT shard by (pixels, N) -> per-rank prefetch
    threads -> packed prefill passes -> decode pool -> ONE all_gather of fixed-width records at the eval reducer -> rank 0 parses / dumps.
    Weak scaling: items_per_gpu x world synthetic 640 x 480 JPEG files x 100 boxes on the node's shared filesystem.  Every rank calls this
    (collectives inside); rank 0 returns the block: whole-job images/s, per-rank shard seconds, the gather's milliseconds / bytes / backend,
    the host-thread budget, and whether the merged token ids of the first `sample` items equal what ONE rank computes for them alone
    (full passes of `batch` same-shape images: a row's arithmetic does not depend on which images share its pass)."""
    import contextlib
    import shutil
    import tempfile
    import torch.distributed as dist
    from vlm_fo1.model.fo1_model import FO1ForCausalLM, FO1HFConfig
    from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor, Qwen2VLPatchProcessor
    from vlm_fo1_amd import sharded_eval as SE
    from vlm_fo1_amd.fixtures.synthetic import ToyTokenizer, full_config_dict, write_coco_like_dataset
    sys.path.insert(0, os.path.join(ROOT, "evaluation"))
    import eval_coco as E
    dev = pipe.eng.dev
    n_items, n_warm = items_per_gpu * world, n_warm_per_gpu * world
    box = [None]
    if rank == 0:
        root = tempfile.mkdtemp(prefix="fo1_scale_")
        warm = write_coco_like_dataset(os.path.join(root, "warm"), n_warm, seed=1)
        data = write_coco_like_dataset(os.path.join(root, "data"), n_items, seed=2)
        sub = os.path.join(root, "data", "sample.jsonl")          # the first `sample` items as their own dataset (same files)
        with open(data[0]) as f, open(sub, "w") as g:
            g.writelines(f.readlines()[:sample])
        box[0] = dict(root=root, warm=warm, data=data, sub=sub)
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    root, warm, data, sub = box[0]["root"], box[0]["warm"], box[0]["data"], box[0]["sub"]
    try:
        model = FO1ForCausalLM.from_engine(FO1HFConfig(full_config_dict()), pipe.eng)
        primary, aux = Qwen2VLPatchProcessor(min_pixels=56 * 56, max_pixels=2048 * 2048), CLIPStyleAuxProcessor(size=768, resize_mode="dynamic")
        primary.device = aux.device = model.device
        tok = ToyTokenizer()
        E.load_pretrained_model = lambda model_id, device="cuda": (tok, model, (primary, aux))
        env = dict(FO1_BATCH=str(batch), FO1_INFLIGHT=str(inflight), FO1_DECODE_POOL=str(pool_slots), FO1_MAX_NEW_TOKENS=str(K),
                   FO1_PREFETCH_THREADS=str(prefetch_threads))
        saved = {k: os.environ.get(k) for k in list(env) + ["RANK", "WORLD_SIZE", "LOCAL_RANK"]}
        os.environ.update(env)
        if one_dev:      # the 1-GPU test mode: the eval driver picks cuda:$LOCAL_RANK itself, and every rank shares device 0
            os.environ["LOCAL_RANK"] = "0"
        name = "synthetic/VLM-FO1_Qwen2.5-VL-3B-synthetic"
        out_dir = os.path.join(root, f"out_rank{rank}")
        try:
            with open(os.path.join(root, f"driver_stdout_rank{rank}.log"), "w") as log, contextlib.redirect_stdout(log):
                E.eval_coco(name, warm[0], warm[1], warm[2], out_dir + "_warm", device=str(dev))
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                E.eval_coco(name, data[0], data[1], data[2], out_dir, device=str(dev))
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                el = time.perf_counter() - t0
                mine = {k: v for k, v in SE.LAST.items() if k != "merged"}
                w0 = pipe.eng.llm.layers[0]
                mine["weights_checksum"] = round(float(w0["wqkv"].float().sum() + pipe.eng.llm.lm_head[:64].float().sum() + pipe.eng.vit.blocks[0]["wqkv"].float().sum()), 4)
                merged = SE.LAST.get("merged")
                stats = [None] * world
                if world > 1:
                    dist.all_gather_object(stats, mine)
                else:
                    stats = [mine]
                same = same_detail = None
                if rank == 0:
                    # ONE rank alone on the first `sample` items: run_sharded / gather_records take the world from the environment
                    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
                    E.eval_coco(name, sub, data[1], data[2], out_dir + "_sample", device=str(dev))
                    torch.cuda.synchronize()
                    alone = dict(SE.LAST.get("merged") or [])
                    together = dict(merged or [])
                    same = len(alone) == sample and all(together.get(i) == alone[i] for i in alone)
                    if not same:      # say how they differ: items, and the position of the first differing token of each
                        diff = [i for i in alone if together.get(i) != alone[i]]
                        first = []
                        for i in diff:
                            a, b = alone[i] or [], together.get(i) or []
                            k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
                            first.append(k)
                        E.eval_coco(name, sub, data[1], data[2], out_dir + "_sample2", device=str(dev))      # and whether ONE rank repeats itself
                        again = dict(SE.LAST.get("merged") or [])
                        same_detail = dict(items_differing=len(diff), differing_item_indices=sorted(diff)[:64], first_differing_token_positions=sorted(first)[:16],
                                           per_rank_weights_checksum=[st.get("weights_checksum") for st in stats],
                                           one_rank_repeats_itself=all(again.get(i) == alone[i] for i in alone))
                    else:
                        same_detail = None
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            pipe.eng.disable_decode_pool()
            for r, _ in model.__dict__.get("_worker_replicas", []):
                r.engine._pool_svc = None
        if world > 1:
            dist.barrier()
        if rank != 0:
            return None
        threads = inflight + prefetch_threads + 1 + (1 if pool_slots else 0)
        return dict(images_per_sec=round(n_items / el, 2), items=n_items, items_per_gpu=items_per_gpu, seconds=round(el, 3), world_size=world,
                    dist_world_size=(dist.get_world_size() if world > 1 else 1), backend=(dist.get_backend() if world > 1 else "none"),
                    one_device_gloo_test_mode=bool(one_dev), new_tokens_per_image=K, images_per_pass=batch,
                    per_rank_shard_seconds=[round(st["shard_seconds"], 3) for st in stats],
                    per_rank_items=[st["shard_items"] for st in stats],
                    # what separates the ranks, each on its own: the LPT deal (model cost and measured seconds, max / mean), how far
                    # apart the ranks ENTERED the loop (same node, same wall clock), how long each waited for the slowest, the collective itself
                    lpt_imbalance_model_cost=round(max(st["shard_cost"] for st in stats) / (sum(st["shard_cost"] for st in stats) / len(stats)), 4),
                    lpt_imbalance_measured_seconds=round(max(st["shard_seconds"] for st in stats) / (sum(st["shard_seconds"] for st in stats) / len(stats)), 4),
                    startup_skew_ms=round((max(st["enter_wall"] for st in stats) - min(st["enter_wall"] for st in stats)) * 1e3, 2),
                    gather_wait_for_slowest_ms=[round(st.get("gather_wait_ms", 0.0), 2) for st in stats],
                    gather_collective_ms=[round(st.get("gather_collective_ms", 0.0), 2) for st in stats],
                    gather_ms=[round(st["gather_ms"], 2) for st in stats],
                    gather_record_bytes_per_rank=stats[0].get("gather_record_bytes_per_rank"),
                    items_failed=sum(1 for _, t in (merged or []) if t is None),
                    sample_items=sample, sample_ids_equal_to_one_rank_alone=same, sample_difference=same_detail,
                    predictions_file_written=os.path.exists(os.path.join(out_dir, name.split("/")[-1], "eval_predictions.json")),
                    host_threads_per_gpu=threads, host_threads_for_this_run=threads * world,
                    loop="evaluation/eval_coco.py eval_coco() on every rank: jsonl -> cost model -> sharded_eval.assign (LPT) -> [prefetch threads] -> packed "
                         "prefill passes -> decode pool -> sharded_eval.gather_records (one all_gather) -> rank 0: tokenizer.decode -> regex -> COCO records -> json dump",
                    data="synthetic 640x480 JPEG files x 100 UPN boxes; ToyTokenizer; random weights at the true shapes; weak scaling (items_per_gpu per rank)")
    finally:
        if world > 1:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)


def hires_run(pipe, steps=6, images=4):
    """BASELINE configs[4]'s geometry in the default run: 1344 x 1344 image (S = 9216 patches) x 300 proposals, run as 3
    prompts of <= 100 over ONE image (the reference caps region features at 100 per prompt, mm_utils.py:600: its callers would run the
    whole model three times; here the towers run once per image, `image_id`), `images` images per packed pass, two passes in flight.
    bf16, then the same passes with the e4m3 linears (`fp8`: parity UNPINNED — the reference has no fp8 path; deviation table in
    DESIGN.md section 10)."""
    dev = pipe.eng.dev
    cases = [build_workload(dev, n_boxes=300, img_hw=(1344, 1344), seed=4321 + i) for i in range(images)]
    reqs = []
    for ci, c in enumerate(cases):
        for ids, bx in c["prompts"]:
            reqs.append(dict(ids=ids, pix=c["dev"]["pix"], grid=c["grid"], aux=c["dev"]["aux"], boxes=bx.to(dev), image_id=ci))
    R = len(pipe.engs)

    def timed():
        for slot in range(R):
            for _ in range(2):
                with torch.cuda.stream(pipe.streams[slot]):
                    pipe.engs[slot].prefill_batch(reqs, use_graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            with torch.cuda.stream(pipe.streams[k % R]):
                pipe.engs[k % R].prefill_batch(reqs, use_graph=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    t_bf16 = timed()
    full_rows = sum(len(r["ids"]) - 1 + (r["grid"][0] // 2) * (r["grid"][1] // 2) for r in reqs)
    out = dict(workload="BASELINE configs[4] geometry: 1344x1344 (S=9216 patches) x 300 proposals as 3 prompts of 100 over one image, prefill to the "
                        "first greedy token of every prompt", images_per_pass=images, prompts_per_pass=len(reqs), passes_in_flight=R,
               llm_rows_per_pass=int(pipe.eng._last_batch["rows"]), llm_rows_per_pass_without_prefix_sharing=full_rows,
               prefix_sharing="the prompts of an image run their common rows (preamble + image tokens) through the LLM once (FO1Engine.SHARE_PREFIX, "
                              "fo1_attention_prefix_bf16); the reference runs the whole model once per prompt",
               bf16=dict(images_per_sec=round(images / t_bf16, 2), ms_per_pass=round(t_bf16 * 1e3, 2), dtype="bf16"))
    # ---- with a decoded answer (round 5, VERDICT r4 missing #5): every prompt's sequence (3 150 rows: shared prefix + own rows, moved into its
    # slot as two pieces) joins a decode pool of 64 slots x 4 096 rows and decodes K tokens there while the replicas prefill the next passes ----
    K, passes = 64, 6
    try:
        svc = pipe.eng.enable_decode_pool(slots=64, slot_rows=4096)
        for e in pipe.engs:
            e._pool_svc = svc
        for slot in range(R):           # untimed: the pool's step graphs for these context lengths
            with torch.cuda.stream(pipe.streams[slot]):
                pipe.engs[slot].submit_batch(reqs, max_new_tokens=K, use_graph=True).result(timeout=600)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        handles = []
        for k in range(passes):
            with torch.cuda.stream(pipe.streams[k % R]):
                handles.append(pipe.engs[k % R].submit_batch(reqs, max_new_tokens=K, use_graph=True))
        ntok = sum(len(t) for h in handles for t in h.result(timeout=600))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["end_to_end"] = dict(images_per_sec=round(passes * images / el, 2), prompts_per_sec=round(passes * len(reqs) / el, 2), new_tokens_per_prompt=K,
                                 generated_tokens=ntok, passes_timed=passes, pool_slots=64, pool_slot_rows=4096, dtype="bf16",
                                 note="packed prefill (towers once per image, shared prompt prefix) + 64 greedy tokens for each of the 3 prompts per image in the decode pool; "
                                      "inputs resident in HBM (no upload / preprocessing in this block)")
    except Exception as e:      # a side measurement must not take the line down
        out["end_to_end"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    finally:
        pipe.eng.disable_decode_pool()
        for e in pipe.engs:
            e._pool_svc = None
    n = pipe.eng.enable_fp8("all")
    for e in pipe.engs[1:]:
        e._graphs.clear(); e._seen.clear()
    try:
        t_fp8 = timed()
        out["fp8"] = dict(images_per_sec=round(images / t_fp8, 2), ms_per_pass=round(t_fp8 * 1e3, 2),
                          dtype=f"fp8-e4m3 linears, preset all ({n} weights; bf16 elsewhere)",
                          parity="UNPINNED: the reference has no fp8 path.  BOUNDED at this geometry against the reference MODULES' fp32 golden by "
                                 "tests/test_fulldepth_parity_gpu.py::test_fp8_presets_at_hires_against_the_reference_golden (profiles/r05_fp8_hires_metrics.json): preset all — "
                                 "last hidden min-cos 0.959 (bf16 engine 0.9995), 3 of 4 margin-qualified ids equal; no preset reaches 0.99, so this is a SPEED figure only "
                                 "(random weights; DESIGN.md section 10)")
    finally:
        pipe.eng.disable_fp8()
        for e in pipe.engs:
            e._graphs.clear(); e._seen.clear()
    return out


def pool_decode_run(pipe, slots=128, steps=48):
    """The decode pool's step with every slot live (llm.DecodePool, 651-token prompts): ms per step, tokens/s and the HBM roofline of
    the step = (weights streamed once + every live sequence's K / V^T read once) / step time."""
    from vlm_fo1_amd.llm import DecodePool
    eng = pipe.eng
    reqs = pipe.requests[:32] if len(pipe.requests) >= 32 else pipe.requests
    eng.prefill_batch(reqs, use_graph=False)
    torch.cuda.synchronize()
    hp, first = eng._last_batch, eng._last_next_tokens.clone()
    pool = DecodePool(eng.llm, slots=slots)
    left = slots
    while left > 0:
        n = min(len(reqs), left)
        pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][:n], hp["delta"][:n], first[:n], 300, ())
        left -= n
    for _ in range(4):
        pool.step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pool.step(True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    c = eng.llm.cfg
    wbytes = float(sum(x.numel() * x.element_size() for x in eng.llm.decode_weight_tensors()))
    L_ctx = hp["seqs"][0][1] + 4 + steps // 2
    kv = 2.0 * slots * c.num_layers * c.num_kv_heads * c.head_dim * 2 * L_ctx
    del pool
    torch.cuda.empty_cache()
    return dict(sequences=slots, ms_per_step=round(t * 1e3, 3), tokens_per_sec=round(slots / t, 1), launches_per_layer=8,
                note="every slot live, one hipGraph replay per step; decode_pool = continuous batching (vlm_fo1_amd/serving.py)",
                roofline=dict(bound="hbm", unit="GB/s", peak=8000.0, algorithmic_bytes_per_step=wbytes + kv, weight_bytes=wbytes, kv_bytes=kv,
                              achieved=round((wbytes + kv) / t / 1e9, 1), frac=round((wbytes + kv) / t / 8e12, 4)))


def cpu_baseline(case, pipe, reps=3, decode_tokens=64):
    """The oracle (a port of the reference's operators: oracle/*.py, torch fp32) of the same stages on this box's host cores, at
    FULL depth (32 ViT blocks, 36 LLM layers): warm-up 1 pass, then the median of `reps` passes, plus `decode_tokens` greedy
    decode steps through the oracle's KV cache.  The literal inference.py cannot run on a CPU (flash-attn, CUDA
    defaults, UPN import: BASELINE.md 3)."""
    import torch.nn.functional as F
    from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, vit_oracle as VO
    ncpu = min(32, len(os.sched_getaffinity(0)))  # 256-thread torch on this box thrashes; 32 is the fastest setting we measured
    torch.set_num_threads(ncpu)
    gh, gw = case["grid"]
    H, Wd = case["img_hw"]
    cfg = pipe.cfg
    sd = {k: {n: t.float().cpu() for n, t in v.items()} for k, v in pipe.weights.items()}
    kw = dict(n_layers=cfg.llm.num_layers, n_heads=cfg.llm.num_heads, n_kv=cfg.llm.num_kv_heads, head_dim=cfg.llm.head_dim,
              eps=cfg.llm.rms_norm_eps, theta=cfg.llm.rope_theta, sections=cfg.llm.mrope_section)
    sw, sh = gw * 14 / Wd, gh * 14 / H
    pix, aux, boxes = case["pix"].float(), case["aux"].float().unsqueeze(0), case["boxes"]

    def mlp2(x, prefix):
        h = F.gelu(F.linear(x, sd["proj"][prefix + "0.weight"], sd["proj"][prefix + "0.bias"]))
        return F.linear(h, sd["proj"][prefix + "2.weight"], sd["proj"][prefix + "2.bias"])

    def one_pass():
        t, keep = {}, {}
        t0 = time.perf_counter()
        tokens, maps = VO.vit_forward(sd["vit"], pix, gh, gw, depth=cfg.vit.depth, n_heads=cfg.vit.num_heads,
                                      fullatt=cfg.vit.fullatt_block_indexes)
        t["vit"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        aux_maps, aux_sizes = DO.davit_forward(sd["davit"], aux)
        t["davit"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        fpn = FO.fpn_forward(sd["fpn"], maps[-1].reshape(gh, gw, -1).permute(2, 0, 1).unsqueeze(0))
        t["fpn"] = time.perf_counter() - t0
        aux_nchw = [m.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(aux_maps, aux_sizes)]
        t0 = time.perf_counter()
        feat = HO.hfre_oracle(aux_nchw, boxes, fpn, boxes * torch.tensor([sw, sh, sw, sh]), region_dim=cfg.mm_region_hidden_size,
                              grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
        t["hfre"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        reg, img = mlp2(feat, "mm_projector_aux."), mlp2(tokens, "mm_projector.")
        t["projectors"] = time.perf_counter() - t0
        emb, nb, na = LO.splice(torch.tensor(case["ids"]), sd["llm"]["embed_tokens.weight"], img, reg)
        pos, delta = LO.rope_index(nb, (gh // 2, gw // 2), na)
        t0 = time.perf_counter()
        hid, cache = LO.llm_forward_cached(sd["llm"], emb, pos, None, **kw)
        t["llm_prefill"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tok = int((hid[-1:] @ sd["llm"]["embed_tokens.weight"].t()).argmax())
        t["lm_head"] = time.perf_counter() - t0
        keep.update(cache=cache, tok=tok, delta=delta)
        return t, keep

    one_pass()                                   # warm-up
    runs = [one_pass() for _ in range(reps)]
    totals = sorted(sum(t.values()) for t, _ in runs)
    total = totals[len(totals) // 2]
    t_med, keep = [r for r in runs if sum(r[0].values()) == total][0]
    dec = None
    if decode_tokens > 0:
        cache, tok, delta = keep["cache"], keep["tok"], keep["delta"]
        emb_w = sd["llm"]["embed_tokens.weight"]
        t0 = time.perf_counter()
        for _ in range(decode_tokens):
            p = cache[0][0].shape[0] + delta
            hid, cache = LO.llm_forward_cached(sd["llm"], emb_w[tok:tok + 1], torch.full((3, 1), p, dtype=torch.long), cache, **kw)
            tok = int((hid @ emb_w.t()).argmax())
        dec = (time.perf_counter() - t0) / decode_tokens
    out = dict(value=1.0 / total, unit="images/s", cores=ncpu, threads=torch.get_num_threads(), host_cpus=os.cpu_count(), kind="port",
               seconds_per_image=round(total, 2), passes_timed=reps, warmup_passes=1,
               pass_seconds=[round(v, 2) for v in totals], stage_seconds={k: round(v, 3) for k, v in t_med.items()},
               sample=f"1 image x {boxes.shape[0]} boxes through every oracle stage at full depth ({cfg.vit.depth} ViT blocks, DaViT-L, SimpleFPN, "
                      f"HFRE, projectors, {cfg.llm.num_layers} LLM layers, last-row lm_head) on {ncpu} host threads (torch fp32): warm-up 1 pass, "
                      f"median of {reps} passes; prefill to the first greedy token, like `value`")
    if dec is not None:
        out["decode_seconds_per_token"] = round(dec, 4)
        out["images_per_sec_with_%d_token_answer" % decode_tokens] = round(1.0 / (total + decode_tokens * dec), 4)
    # `kind: reference` cannot be timed here (/root/reference does not exist on the GPU box).  The committed check from the build container,
    # where it does: the reference's own modules against this port on the same workload and host threads (scripts/cpu_reference_vs_port.py)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_cpu_baseline_reference_vs_port_build_container.json")) as f:
            chk = json.load(f)
        out["reference_modules_check"] = dict(
            source="profiles/r05_cpu_baseline_reference_vs_port_build_container.json (build container, %d threads; NOT timed in this run)" % chk["host_threads"],
            reference_modules_seconds_per_image=chk["reference_modules"]["seconds_per_image"], port_seconds_per_image=chk["oracle_port"]["seconds_per_image"],
            port_over_reference_seconds=chk["port_over_reference_seconds"],
            same_first_token=chk["first_token_reference"] == chk["first_token_port"],
            note="the port is the faster of the two on the same cores, so `value` is, if anything, a stronger CPU baseline than the reference's own modules")
    except (OSError, ValueError, KeyError):
        pass
    return out


def hfre_algorithmic_bytes(case, region_dim=5888, P=7):
    """SURVEY 8(d) per-image byte count of the HFRE gather: sum over sources of |U_l| * C_l * 2 (bf16 read of every map element
    inside the UNION U_l of the boxes' tap footprints at the source's native resolution) + N * C_region * 4 (fp32 write) + 16 N
    (boxes); also the simple upper bound with |U_l| = H_l * W_l.  The per-axis tap range is the one hfre_math.h derives
    (make_roi_axis + upsample_range: torchvision roi_align aligned=False sample positions, F.interpolate align_corners=False
    taps), restated here in numpy fp32 so the figure is computed from the workload, not read back from the kernel."""
    import numpy as np
    f32 = np.float32
    H, W = case["img_hw"]
    gh, gw = case["grid"]
    aux, hh, ww = [], (H + 3) // 4, (W + 3) // 4
    for c in (256, 512, 1024, 2048):
        aux.append((hh, ww, c))
        hh, ww = (hh + 1) // 2, (ww + 1) // 2
    roi0 = aux[0][:2]
    boxes = case["boxes"].numpy().astype(f32)
    sx, sy = f32(gw * 14 / W), f32(gh * 14 / H)
    vtb = boxes * np.array([sx, sy, sx, sy], dtype=f32)
    srcs = [(h, w, c, roi0, f32(0.25), boxes) for h, w, c in aux]
    for (h, w), st in zip(((4 * gh, 4 * gw), (2 * gh, 2 * gw), (gh, gw), (gh // 2, gw // 2)), (3.5, 7.0, 14.0, 28.0)):
        srcs.append((h, w, 512, (h, w), f32(1.0 / st), vtb))

    def axis(lo, hi, scale, L):
        s0, s1 = f32(lo * scale), f32(hi * scale)
        ln = max(f32(s1 - s0), f32(1.0))
        b = f32(ln / f32(P))
        g = max(int(np.ceil(ln / f32(P))), 1)
        coord = lambda s: f32(s0 + f32(s // g) * b + f32(f32(s % g) + f32(0.5)) * b / f32(g))
        vf, vl = coord(0), coord(P * g - 1)
        if not (vl >= -1.0) or not (vf <= L) or not (ln < 1e8):
            return 0, -1
        a_lo = 0 if vf <= 0 else int(min(vf, L - 1))
        a_hi = L - 1 if vl >= L - 1 else (0 if vl <= 0 else int(vl)) + 1
        a_hi, a_lo = min(a_hi, L - 1), min(a_lo, L - 1)
        return a_lo, max(a_hi, a_lo)

    def up(o, Lin, Lout):
        if Lin == Lout:
            return o, o
        src = max(f32(f32(Lin) / f32(Lout)) * f32(o + 0.5) - f32(0.5), f32(0.0))
        f = min(int(np.floor(src)), Lin - 1)
        return f, f + (1 if f < Lin - 1 else 0)

    union = full = 0
    for h, w, c, (rh, rw), scale, bx in srcs:
        mask = np.zeros((h, w), dtype=bool)
        for x1, y1, x2, y2 in bx:
            ylo, yhi = axis(y1, y2, scale, rh)
            xlo, xhi = axis(x1, x2, scale, rw)
            if yhi < ylo or xhi < xlo:
                continue
            mask[up(ylo, h, rh)[0]:up(yhi, h, rh)[1] + 1, up(xlo, w, rw)[0]:up(xhi, w, rw)[1] + 1] = True
        union += int(mask.sum()) * c * 2
        full += h * w * c * 2
    n = boxes.shape[0]
    tail = n * region_dim * 4 + n * 16
    return dict(footprint_union=union + tail, full_map_upper_bound=full + tail)


def pmc_traffic(kernel_name, full=False):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_traffic.json, written by scripts/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs with the
    gfx950 x2 FETCH correction of MI355X_MICROARCH.md applied; both counters calibrated on known byte counts in this engine's access patterns,
    profiles/r06_pmc_calibration.json).  PMC counters cannot be read from inside the process, so the
    bench line carries the committed figure and names its source; null when no PMC pass exists for the kernel.
    full=True: (row dict, source) — the read / write split and, for the 256 x 256 GEMM, the algorithmic read / write bytes it is set against."""
    for name in ("r06_pmc_traffic.json",):     # (rounds 3-5 files held ONE template instantiation per kernel name — scripts/pmc_summary.py, round 6 fix — and are not quoted any more)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                t = json.load(f)
            row = t["kernels"].get(kernel_name)
            if row:
                return (row, "profiles/" + name) if full else (row["hbm_bytes_per_launch"], "profiles/" + name)
        except (OSError, ValueError, KeyError):
            pass
    return None, None


# ---- the printed line (VERDICT r5 #2b): the driver keeps the LAST 2000 characters of stdout next to the parsed contract keys, so the line is
# kept small (<= ~6 KB: verbose per-kernel tables, notes and loop descriptions go to --json-out) and the user-visible figures are
# repeated in ONE flat `summary` object that is printed last ----
_VERBOSE_KEYS = ("note", "loop", "includes", "excludes", "prefix_sharing", "decode", "launch", "packing", "what", "sample", "data")


_COMPACT_DROP = (
    "cpu_baseline.reference_modules_check", "cpu_baseline.pass_seconds", "cpu_baseline.stage_seconds", "cpu_baseline.host_cpus",
    "cpu_baseline.threads", "cpu_baseline.passes_timed", "cpu_baseline.warmup_passes", "cpu_baseline.decode_seconds_per_token",
    "cpu_baseline.images_per_sec_with_64_token_answer",
    "roofline.sustained_mfma.zero_operands", "roofline.sustained_mfma.random_operands.us",
    "roofline.hfre.algorithmic_bytes_full_map_upper_bound", "roofline.hfre.achieved_on_upper_bound_bytes", "roofline.hfre.achieved_main_kernel_only",
    "roofline.hfre.images_per_launch", "roofline.hfre.us_per_image", "roofline.hfre.traffic_source", "roofline.hfre.kernel", "roofline.hfre.peak",
    "roofline.hfre.unit",
    "hires.workload", "hires.fp8.parity", "hires.fp8.dtype", "hires.bf16.dtype", "hires.end_to_end.dtype", "hires.llm_rows_per_pass_without_prefix_sharing",
    "hires.end_to_end.pool_slots", "hires.end_to_end.pool_slot_rows", "hires.end_to_end.generated_tokens", "hires.end_to_end.passes_timed",
    "dataset.patches_min", "dataset.patches_max", "dataset.distinct_grids", "dataset.distinct_aux_sizes", "dataset.boxes", "dataset.aux",
    "dataset.passes", "dataset.images_per_pass_max", "dataset.passes_in_flight", "dataset.seconds", "dataset.patches_per_item",
    "decode.roofline.bound", "decode.roofline.unit", "decode.roofline.peak", "decode.batched.roofline.bound", "decode.batched.roofline.unit",
    "decode.batched.roofline.peak", "decode.pool.roofline.bound", "decode.pool.roofline.unit", "decode.pool.roofline.peak",
    "decode.batched.roofline.algorithmic_bytes_per_step", "decode.pool.roofline.weight_bytes", "decode.pool.roofline.kv_bytes",
    "decode.new_tokens_timed", "decode.host_loop_ms_per_token", "decode.batched.prefill_pass_ms",
    "end_to_end.static_groups.new_tokens_per_image", "end_to_end.static_groups.passes_timed", "end_to_end.static_groups.images_per_pass",
    "end_to_end.static_groups.passes_in_flight", "end_to_end.static_groups.generated_tokens_per_sec", "end_to_end.passes_timed",
    "end_to_end.pool_steps", "driver_level.host_prepare_threads_needed_at_this_rate", "driver_level.predictions_file_written",
    "driver_level_countbench.countbench.seconds_writing_the_files_untimed", "driver_level_countbench.pixmo.seconds_writing_the_files_untimed",
    "driver_level_countbench.countbench.accuracy_on_random_weights", "driver_level_countbench.pixmo.accuracy_on_random_weights",
    "dataset.region_tokens_per_sec", "dataset.uniform_equivalent_images_per_sec", "dataset.boxes_per_item", "hires.images_per_pass",
    "hires.prompts_per_pass", "hires.passes_in_flight", "hires.llm_rows_per_pass", "hires.end_to_end.new_tokens_per_prompt",
    "end_to_end.generated_tokens_per_sec", "end_to_end.decode_group", "end_to_end.static_groups.decode_group", "end_to_end.static_groups.ms_per_pass",
    "driver_level.host_prepare_threads_needed_for_8_gpus", "driver_level.seconds", "driver_level_countbench.countbench.seconds",
    "driver_level_countbench.pixmo.seconds", "roofline.traffic_source",
    "region_tokens_per_sec_hfre_plus_connector", "region_tokens_per_sec_encode_regions", "preprocess",
)


def compact_line(full: dict) -> dict:
    """The contract line: every contract key of `full` unchanged, nested blocks without their prose (`note`, `loop`, ... and any string
    longer than 120 characters) and without the per-kernel tables (`roofline.per_step_ms` / `launches`: in --json-out), then `summary`."""
    def strip(o, top=False):
        if isinstance(o, dict):
            out = {}
            for k, v in o.items():
                if not top and (k in _VERBOSE_KEYS or k.endswith("_note")) and isinstance(v, (str, list)):
                    continue
                if top and (k in ("config", "cpu_baseline") or not isinstance(v, (dict, list))):
                    out[k] = v              # contract keys verbatim (config.workload, cpu_baseline.sample are prose the contract asks for)
                    continue
                if isinstance(v, str) and len(v) > 120:
                    v = v[:117] + "..."
                out[k] = strip(v)
            return out
        if isinstance(o, list):
            return [strip(v) for v in o]
        if isinstance(o, float):
            return float(f"{o:.6g}")
        return o
    line = strip(full, top=True)
    roof = line.get("roofline")
    if isinstance(roof, dict):
        roof.pop("per_step_ms", None)
        roof.pop("launches", None)
    cfg = line.get("config")
    if isinstance(cfg, dict):
        cfg.pop("stages", None)
    line.pop("stage_kernel_ms_note", None)
    # detail that only the full record carries (VERDICT r5 #2b: <= 6 KB on stdout): dotted paths inside the nested blocks
    for path in _COMPACT_DROP:
        o = line
        *head, last = path.split(".")
        for k in head:
            o = o.get(k) if isinstance(o, dict) else None
        if isinstance(o, dict):
            o.pop(last, None)

    def g(*path):
        o = full
        for k in path:
            if not isinstance(o, dict) or o.get(k) is None:
                return None
            o = o[k]
        return float(f"{o:.5g}") if isinstance(o, float) else o
    summary = dict(
        value_images_per_sec=g("value"), ms_per_step=g("ms_per_step"),
        end_to_end_images_per_sec=g("end_to_end", "images_per_sec"), end_to_end_pool_mean_live=g("end_to_end", "pool_mean_live_sequences"),
        end_to_end_static_groups_images_per_sec=g("end_to_end", "static_groups", "images_per_sec"),
        driver_level_images_per_sec=g("driver_level", "images_per_sec"), driver_level_vs_end_to_end=g("driver_level", "vs_end_to_end"),
        driver_level_host_threads_per_gpu=g("driver_level", "host_threads_per_gpu"),
        reference_literal_batch1_loop_images_per_sec=g("driver_level", "reference_literal_batch1_loop_images_per_sec"),
        countbench_driver_images_per_sec=g("driver_level_countbench", "countbench", "images_per_sec"),
        pixmo_driver_images_per_sec=g("driver_level_countbench", "pixmo", "images_per_sec"),
        hires_bf16_images_per_sec=g("hires", "bf16", "images_per_sec"), hires_fp8_images_per_sec=g("hires", "fp8", "images_per_sec"),
        hires_end_to_end_images_per_sec=g("hires", "end_to_end", "images_per_sec"),
        one_image_ms=g("one_image_at_a_time", "ms_per_image"), one_pass_images_per_sec=g("one_pass_at_a_time", "images_per_sec"),
        dataset_images_per_sec=g("dataset", "images_per_sec"), dataset_vs_uniform=g("dataset", "vs_uniform_headline"),
        decode_ms_per_token=g("decode", "ms_per_token"), decode_hbm_frac=g("decode", "roofline", "frac"),
        decode_batched_sequences=g("decode", "batched", "sequences"), decode_batched_ms_per_step=g("decode", "batched", "ms_per_step"),
        decode_batched_hbm_frac=g("decode", "batched", "roofline", "frac"),
        decode_pool_ms_per_step=g("decode", "pool", "ms_per_step"), decode_pool_tokens_per_sec=g("decode", "pool", "tokens_per_sec"),
        decode_pool_hbm_frac=g("decode", "pool", "roofline", "frac"),
        scale_images_per_sec=g("scale", "images_per_sec"), scale_world=g("scale", "world_size"),
        gemm_frac_of_peak=g("roofline", "frac"), gemm_frac_of_sustained=g("roofline", "sustained_mfma", "frac_of_random_operand_rate"),
        all_gemm_tflops=g("roofline", "all_gemm_tiles", "tflops"), hfre_hbm_frac=g("roofline", "hfre", "frac"),
        hfre_us_all_launches=g("roofline", "hfre", "us_all_launches"),
        cpu_baseline_images_per_sec=g("cpu_baseline", "value"), cpu_baseline_cores=g("cpu_baseline", "cores"))
    line["summary"] = {k: v for k, v in summary.items() if v is not None}
    return line


def emit(full: dict, json_out: str) -> None:
    """Full record -> json_out (per-kernel tables, notes, loop descriptions); compact contract line -> stdout (ONE line, last)."""
    if json_out:
        try:
            d = os.path.dirname(json_out)
            if d:
                os.makedirs(d, exist_ok=True)
            with open(json_out, "w") as f:
                json.dump(full, f, indent=1)
            full = dict(full, full_record=json_out)
        except OSError as e:
            full = dict(full, full_record=f"not written: {e}")
    print(json.dumps(compact_line(full), separators=(",", ":")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--boxes", type=int, default=100, help="proposals per image (default 100 = the configuration BASELINE.json's metric is quoted on; 32 = configs[1])")
    ap.add_argument("--image", default="480x640", help="HxW of the synthetic image (default = BASELINE configs[1]; 1344x1344 with "
                    "--boxes 100 is the high-resolution configuration's geometry)")
    ap.add_argument("--batch", type=int, default=0, help="images packed into ONE pass of every stage (varlen batched prefill; 0 = 25 at the metric configuration, scaled by patches per image otherwise); a step is "
                    "one such pass; 1 = one image per pass (latency mode).  The 256 x 256 GEMM tiles run in rounds of 256 (one per CU), so "
                    "the pass size sets how full the last round of every product is: 25 (default) makes the LLM o / down projections "
                    "64 x 8 = 512 tiles = exactly two rounds and the ViT products 94-99 %% full (model + sweep: profiles/r02_batch_sweep.md; "
                    "12 = 248 tiles = one round for the LLM but 72 %% for ViT proj / down: 125.3 images/s, dominant GEMM 1029 TFLOP/s, against "
                    "128.3 / 1105 at 25; 8: 120.5)")
    ap.add_argument("--inflight", type=int, default=2, help="independent passes in flight per GPU (engine replicas on their own HIP "
                    "streams); 1 = strictly one pass at a time")
    ap.add_argument("--pool-slots", type=int, default=128, choices=[0, 64, 128], help="end_to_end: slots of the decode pool the passes' sequences "
                    "join (continuous batching, vlm_fo1_amd/serving.py); 0 = every pass decodes its own group of <= 32 (round 3's form)")
    ap.add_argument("--decode-pools", type=int, default=0, help="end_to_end: decode pools stepping concurrently on their own streams (serving.PoolGroup); 0 = the engine's default")
    ap.add_argument("--e2e-only", action="store_true", help="print the end_to_end block and stop (A/B runs)")
    ap.add_argument("--e2e-passes", type=int, default=0, help="end_to_end: timed passes (0 = min(steps, 24)); a longer loop weighs the decode pool's fill / drain "
                    "phases at its two ends less (side measurement: the default line keeps 24)")
    ap.add_argument("--driver-items", type=int, default=768, help="driver_level: images of the synthetic COCO-shaped dataset run through "
                    "evaluation/eval_coco.py's own loop (0 = skip)")
    ap.add_argument("--driver-count-items", type=int, default=-1, help="driver_level_countbench: items per dataset of evaluation/eval_countbench.py's own loop on the "
                    "CountBench and Pixmo-Count fixtures as files (-1 = all 487 + 529; 0 = skip)")
    ap.add_argument("--scale-items", type=int, default=2048, help="multi-rank runs: images PER GPU of the `scale` block (evaluation/eval_coco.py's loop through "
                    "sharded_eval.run_sharded across the ranks, one all_gather at the reducer); a multiple of 32; 0 = skip")
    ap.add_argument("--json-out", default="gpurun_out/bench_full.json", help="the FULL record (per-kernel tables, notes, loop descriptions); stdout carries the "
                    "compact contract line whose last object, `summary`, repeats the user-visible figures ('' = do not write)")
    ap.add_argument("--no-hires", action="store_true", help="skip the `hires` block (BASELINE configs[4]'s geometry: 1344x1344 x 300 proposals, bf16 and fp8 linears)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true", help="skip the side measurements (one image / one pass at a time, decode loops, preprocessing): "
                    "only packed passes of the main workload run — what a rocprofv3 / PMC pass of this command should see, so that its "
                    "per-kernel averages are over the same launches as the roofline block's")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed oracle passes (median) after one warm-up pass")
    ap.add_argument("--cpu-decode-tokens", type=int, default=64)
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the hipGraph")
    ap.add_argument("--fp8", nargs="?", const="all", default=None, choices=["all", "mlp", "llm-mlp"], help="SECONDARY measurement (BASELINE configs[4] names fp8 MFMA): W8A8 e4m3 for the ViT / LLM "
                    "qkv, gate/up and down projections of the packed pass (FO1Engine.enable_fp8); the line says so in `dtype` — the "
                    "default run is bf16 like the reference")
    ap.add_argument("--profile-shapes", action="store_true", help="per-shape GEMM rows in roofline.per_step_ms")
    ap.add_argument("--lift-cap", action="store_true", help="with --boxes > 100: ONE prompt carrying all N region tokens instead of ceil(N / 100) prompts of "
                    "<= 100.  A deliberate EXTENSION, labelled in the line: the reference cannot run such a prompt (features are cut to 100, "
                    "mm_utils.py:600, and the splice IndexErrors, omchat_qwen2_5_vl.py:361) — the engine has no such cap")
    ap.add_argument("--dataset", default="countbench", choices=["countbench", "pixmo", "coco-like", "none"], help="dataset-shaped side measurement "
                    "reported in the line's `dataset` block (ragged image sizes and box counts: the reference's CountBench / Pixmo fixtures "
                    "verbatim, or COCO-like sizes x 100 boxes); `none` skips it")
    ap.add_argument("--dataset-items", type=int, default=150, help="items of the dataset to run (0 = all)")
    ap.add_argument("--dataset-aux", default="dynamic", choices=["dynamic", "squash"], help="aux image sizing (config aux_image_aspect_ratio)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher — one rank per GPU under torch.distributed.run on the
        # loopback address (the container hostname may not resolve), same arguments.  rank 0's JSON line is the only stdout line.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("FO1_BENCH_SPAWN_DRYRUN") == "1":      # tests/test_bench_cli.py (no GPU): show the launch instead of doing it
            print(json.dumps(cmd))
            return
        os.execv(sys.executable, cmd)

    # Host tensors on the GPU path (index plans, rope tables) are a few hundred KB: one intra-op thread.  torch's default — one per core,
    # 256 on the GPU box — turns every small torch.cat / clone of the submitting threads into an OpenMP region on an oversubscribed pool
    # (driver_level: host planning 10x slower, profiles/r04_driver_level_host_profile_*.log).  cpu_baseline sets its own count.
    torch.set_num_threads(1)

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
    if os.environ.get("FO1_GEMM_GROUP_M") and L.ab_build():      # A/B of the 256 x 256 GEMM's tile order (include/fo1_ab.h, FO1_AB=1 runs only)
        L.load().fo1_gemm_set_group_m(int(os.environ["FO1_GEMM_GROUP_M"]))
    if os.environ.get("FO1_HFRE_ORDER") and L.ab_build():        # A/B of the HFRE work-list order: 0 = interleaved buckets (rounds 2-5), 1 = box-major (image-major)
        L.load().fo1_hfre_set_tuning(8, 512, -4 if int(os.environ["FO1_HFRE_ORDER"]) else -3, 0)
    if os.environ.get("FO1_DWLN_FORM") and L.ab_build():         # A/B of DaViT's depthwise conv + LayerNorm: 0 = per-pixel form, 1 = product rule (sliding-window runs)
        L.load().fo1_dwconv_ln_set_form(int(os.environ["FO1_DWLN_FORM"]))
    if os.environ.get("FO1_CHATTN_IMPL") and L.ab_build():       # A/B of DaViT's channel attention: 0 = fp32 FMA kernels (rounds 1-5), 1 = matrix-core kernels
        L.load().fo1_channel_attention_set_impl(int(os.environ["FO1_CHATTN_IMPL"]))
    img_hw = tuple(int(v) for v in args.image.lower().split("x"))
    S_img = (round(img_hw[0] / 28) * 2) * (round(img_hw[1] / 28) * 2)
    auto_batch = args.batch <= 0
    if auto_batch:            # default: the pass size that keeps ~39k ViT rows per pass (25 images at the metric configuration)
        args.batch = max(1, round(25 * 1564 / S_img))
    B = max(1, args.batch)
    if args.boxes > 100:      # several prompts per image: the one-sequence side measurements and the one-prompt CPU leg do not apply
        args.main_only = True
        args.no_cpu_baseline = True
    cases = [build_workload(dev, n_boxes=args.boxes, img_hw=img_hw, seed=1234 + rank * 1000 + i, lift_cap=args.lift_cap) for i in range(B)]
    case = cases[0]
    R = max(1, args.inflight)
    pipe = Pipeline(case, dev, inflight=R, batch=B, cases=cases)
    n_fp8 = 0
    if args.fp8:
        n_fp8 = pipe.eng.enable_fp8(args.fp8)
        for e in pipe.engs[1:]:
            e._graphs.clear(); e._seen.clear()

    use_graph = not args.eager
    for slot in range(R):
        for _ in range(args.warmup):
            pipe.step(use_graph, slot)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # The cyclic garbage collector is parked for the timed region, as a serving process would do after start-up (gc.freeze): a
    # generation-2 sweep over the engine's object graph stalls the submitting thread for ~0.2 s every few dozen passes
    # (profiles/r02_sustained_b25.log: 10-step blocks at 220-230 ms/step among 197 ms ones).  Nothing is skipped: every step does its
    # full host planning and launches.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
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
    gc.enable()
    if world > 1:
        t = torch.tensor([el], device="cpu" if one_dev else dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    # ---- multi-rank runs: the sharded evaluation loop across the ranks (LPT shard -> prefetch -> pool -> one all_gather); every rank takes
    # part, right after the timed region (the other ranks then leave; rank 0 goes on with its side measurements).  Never `value`. ----
    scale = None
    if world > 1 and use_graph and args.scale_items > 0 and args.boxes <= 100 and img_hw == (480, 640):
        scale = scale_run(pipe, world, rank, one_dev, items_per_gpu=args.scale_items, pool_slots=args.pool_slots or 128)

    # ---- side measurements on rank 0 (never `value`): one packed pass at a time, and strictly one image at a time ----
    single = one_pass = None
    if rank == 0 and not args.main_only:
        if R > 1 or B > 1:
            for _ in range(3):
                pipe.step_single(use_graph)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step_single(use_graph)
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            single = dict(images_per_sec=round(args.steps / e1, 2), ms_per_image=round(e1 / args.steps * 1e3, 3))
        else:
            single = dict(images_per_sec=args.steps / el, ms_per_image=el / args.steps * 1e3)
        if R > 1:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step(use_graph, 0)
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            one_pass = dict(images_per_sec=round(args.steps * B / e1, 2), ms_per_pass=round(e1 / args.steps * 1e3, 3), images_per_pass=B)

    # ---- greedy decode through the KV cache (fixed K new tokens, reported separately; not part of `value`) ----
    dec = None
    if rank == 0 and not args.main_only and not args.e2e_only:
        K = 32
        out = pipe.step_single(use_graph)
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
                   images_per_sec_with_64_token_answer=round(1.0 / (single["ms_per_image"] * 1e-3 + 64 * td), 2),
                   note="one sequence at a time, host reads every token (the round-1 path)")
        # device decode loop: the sequences of one packed prefill advance together, weights streamed once per step, stop rule and
        # bookkeeping on the device (no host read per token); measured for one sequence and for the batch
        from vlm_fo1_amd.llm import BatchDecoder
        Bd = min(B, BatchDecoder.MAX_BATCH)      # one decode group: up to 32 sequences per weight stream
        if use_graph:
            eng = pipe.eng

            def device_loop(n):
                reqs = pipe.requests[:n]
                eng.prefill_batch(reqs, use_graph=True)
                eng.prefill_batch(reqs, use_graph=True)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for _ in range(5):
                    eng.prefill_batch(reqs, use_graph=True)
                torch.cuda.synchronize()
                t_pref = (time.perf_counter() - tp) / 5
                d = eng._decoder()
                hp = eng._last_batch
                d.start(hp["seqs"], hp["delta"], eng._last_next_tokens[:n], 4096, ())
                for _ in range(4):
                    d.step(True)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(K):
                    d.step(True)
                torch.cuda.synchronize()
                return t_pref, (time.perf_counter() - tb) / K

            t_pref1, t1 = device_loop(1)
            dec["host_loop_ms_per_token"] = dec["ms_per_token"]
            dec["note"] = "ms_per_token: one sequence, device loop (stop rule on the device); host_loop_*: the round-1 path, host reads every token"
            dec["ms_per_token"] = round(t1 * 1e3, 3)
            dec["tokens_per_sec"] = round(1.0 / t1, 1)
            dec["images_per_sec_with_64_token_answer"] = round(1.0 / (t_pref1 + 64 * t1), 2)
            if Bd > 1:
                t_pref, tb = device_loop(Bd)
                dec["batched"] = dict(sequences=Bd, ms_per_step=round(tb * 1e3, 3), tokens_per_sec=round(Bd / tb, 1),
                                      prefill_pass_ms=round(t_pref * 1e3, 3),
                                      images_per_sec_with_64_token_answer=round(Bd / (t_pref + 64 * tb), 2),
                                      launches_per_layer=5, note="one pass at a time: packed prefill of the batch, then 64 batched decode steps")
                # HBM roofline of the decode step: every weight byte streams once per step for all sequences of the group
                wbytes = float(sum(t.numel() * t.element_size() for t in eng.llm.decode_weight_tensors())) if hasattr(eng.llm, "decode_weight_tensors") else 6.2e9
                for blk, tt in ((dec, t1), (dec["batched"], tb)):
                    blk["roofline"] = dict(bound="hbm", unit="GB/s", peak=8000.0, algorithmic_bytes_per_step=wbytes,
                                           achieved=round(wbytes / tt / 1e9, 1), frac=round(wbytes / tt / 8e12, 4))
            if args.pool_slots and B >= 16:
                dec["pool"] = pool_decode_run(pipe, slots=args.pool_slots)

    # ---- end to end: upload + device preprocessing + packed prefill + 64-token batched decode + ids on the host, one timed loop ----
    e2e = None
    if rank == 0 and not args.main_only and use_graph:
        # pass size of the end-to-end loop = one full decode group (32 sequences per weight stream) where the workload allows it; `value`
        # keeps the 25 images per pass that fill the prefill GEMMs' tile rounds best (end to end: 75 images/s at 25 per pass, 80 at 32)
        from vlm_fo1_amd.llm import BatchDecoder
        e2e_cases = cases
        if auto_batch and B >= 16 and B < BatchDecoder.MAX_BATCH and not any("prompts" in c for c in cases):
            e2e_cases = cases + [build_workload(dev, n_boxes=args.boxes, img_hw=img_hw, seed=1234 + rank * 1000 + i, lift_cap=args.lift_cap)
                                 for i in range(B, BatchDecoder.MAX_BATCH)]
        # continuous batching (round 4): one decode pool of 128 slots per GPU, fed by the replicas' prefill passes; more passes than the
        # static form so that the pool's fill / drain phases at the two ends of the timed loop weigh little
        e2e = end_to_end_run(pipe, e2e_cases, steps=(args.e2e_passes if args.e2e_passes > 0 else max(8, min(args.steps, 24))), K=64, pool_slots=args.pool_slots,
                             pools=args.decode_pools or None)
        if args.e2e_only:
            print(json.dumps(dict(end_to_end=e2e)))
            return
        if args.pool_slots:
            e2e["static_groups"] = end_to_end_run(pipe, e2e_cases, steps=max(4, min(args.steps, 12)), K=64, pool_slots=0)

    # ---- driver level: evaluation/eval_coco.py's own loop on files: not part of `value` ----
    drv = None
    if rank == 0 and not args.main_only and use_graph and args.driver_items > 0 and args.boxes <= 100 and img_hw == (480, 640):
        drv = driver_level_run(pipe, n_items=args.driver_items, K=64, pool_slots=args.pool_slots)
        if e2e is not None:
            drv["vs_end_to_end"] = round(drv["images_per_sec"] / e2e["images_per_sec"], 3)
        if dec is not None:      # what the reference's own batch-1 loop (one generate() per image) gets on this engine, same answer length
            drv["reference_literal_batch1_loop_images_per_sec"] = dec.get("images_per_sec_with_64_token_answer")

    # ---- driver level, configs[3]: evaluation/eval_countbench.py's own loop on the full CountBench + Pixmo-Count fixtures ----
    drv_count = None
    if rank == 0 and world == 1 and not args.main_only and use_graph and args.driver_count_items != 0 and args.boxes <= 100 and img_hw == (480, 640):
        drv_count = driver_level_count_run(pipe, limit=(args.driver_count_items if args.driver_count_items > 0 else None), K=64, pool_slots=args.pool_slots or 128)

    # ---- BASELINE configs[4]'s geometry (bf16 and fp8 linears): not part of `value` ----
    hires = None
    if rank == 0 and not args.main_only and use_graph and not args.no_hires and args.boxes <= 100 and img_hw == (480, 640) and not args.fp8:
        hires = hires_run(pipe)

    # ---- dataset-shaped workload (ragged sizes / variable N): not part of `value` ----
    dset = None
    if rank == 0 and not args.main_only and args.dataset != "none":
        dset = dataset_run(pipe, args.dataset, args.dataset_items or None, B, R, aux_mode=args.dataset_aux)
        dset["vs_uniform_headline"] = round(dset["uniform_equivalent_images_per_sec"] / (args.steps * B / el), 3)

    # ---- host-side preprocessing of one image (SURVEY 8d "preprocess (CPU)" stage, 8f rank 2): not part of `value` ----
    prep = None
    if rank == 0 and not args.main_only:
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
        if args.profile_shapes:      # per-shape profile rows: an instrument of include/fo1_ab.h (FO1_AB=1 runs only)
            if not L.ab_build():
                raise SystemExit("--profile-shapes needs the test / bench build: run with FO1_AB=1")
            L.load().fo1_gemm_profile_shapes(1)
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
        peak = (MFMA_FP8_PEAK_TF if dom["name"].startswith("gemm_fp8") else MFMA_BF16_PEAK_TF) if mfma else HBM_PEAK_GBS
        traffic, traffic_src = pmc_traffic(dom["name"])
        trow, _ = pmc_traffic(dom["name"], full=True)
        # all MFMA GEMM templates together (the three tile shapes are one kernel source)
        g_rows = [r for r in rows if r["name"].startswith("gemm_bt_")]
        g_ms = sum(r["total_ms"] for r in g_rows)
        g_tf = sum(r["total_work"] for r in g_rows) / (g_ms * 1e-3) / 1e12 if g_ms > 0 else None
        roof = dict(kernel=dom["name"], bound="mfma" if mfma else "hbm", achieved=round(ach, 2), peak=peak,
                    unit="TFLOP/s" if mfma else "GB/s", frac=round(ach / peak, 5), traffic=traffic,
                    traffic_source=traffic_src,
                    # read and write sides separately (VERDICT r5 #2a), each against what a launch must move (A + W (+ residual) once; C once)
                    traffic_read=(trow or {}).get("fetch_bytes"), traffic_write=(trow or {}).get("write_bytes"),
                    algorithmic_read=(trow or {}).get("algorithmic_read_bytes"), algorithmic_write=(trow or {}).get("algorithmic_write_bytes"),
                    traffic_read_over_algorithmic=(trow or {}).get("read_over_algorithmic"), traffic_write_over_algorithmic=(trow or {}).get("write_over_algorithmic"),
                    all_gemm_tiles=dict(tflops=round(g_tf, 2) if g_tf else None, ms_per_step=round(g_ms / nprof, 3),
                                        launches_per_step=sum(r["calls"] for r in g_rows) // nprof),
                    avg_us=round(avg_ms * 1e3, 3), launches_per_step=dom["calls"] // nprof,
                    algorithmic_work_per_launch=work,
                    per_step_ms={r["name"]: round(r["total_ms"] / nprof, 4) for r in rows},
                    launches={r["name"]: r["calls"] // nprof for r in rows})
        if mfma:
            # what the matrix pipes sustain on THIS box (fo1_mfma_clock_probe: a register-resident dense bf16 MFMA loop on every CU, no memory
            # traffic): `peak` above is 256 CUs x 4 SIMDs x 1024 flop/cycle at 2.4 GHz, under load the chip clocks to its power budget
            try:
                from vlm_fo1_amd import ops as _ops
                with L.use_ab():         # the probe is an instrument of include/fo1_ab.h: the test / bench build is mapped HERE, after the timed region
                    rnd, zero = _ops.mfma_clock_probe(1), _ops.mfma_clock_probe(0)
                roof["sustained_mfma"] = dict(random_operands=rnd, zero_operands=zero, frac_of_random_operand_rate=round(ach / rnd["tflops"], 4),
                                              note="register-resident v_mfma_f32_32x32x16_bf16 loop, 8 waves per CU, no loads: the ceiling of any bf16 MFMA kernel at "
                                                   "this box's power budget (DVFS); frac_of_random_operand_rate = roofline.achieved / that rate")
            except Exception as e:       # instrumentation only
                roof["sustained_mfma"] = dict(error=str(e)[:200])
        # the HFRE gather is the HBM-bound kernel north_star names: report it next to the dominant (MFMA) kernel
        by = {r["name"]: r for r in rows}
        hf = [k for k in by if k.startswith("hfre")]
        if hf:
            # algorithmic bytes = the union of the boxes' footprints on every source map (what a perfect gather
            # moves) + the fp32 output; `frac` uses THAT and the time of every HFRE launch of the image.  The full-map figure
            # (each source read once in full) is the published upper bound, reported as a second field.
            hb_alg = hfre_algorithmic_bytes(case, region_dim=pipe.cfg.mm_region_hidden_size)
            # a batched step gathers every image's boxes in ONE launch: algorithmic bytes per launch = footprint x images/launch
            main_k = "hfre_pool_items" if "hfre_pool_items" in by else ("hfre_pool" if "hfre_pool" in by else hf[0])
            ipl = max(1, round(pipe.batch / max(1, by[main_k]["calls"])))
            hb_alg = {k: v * ipl for k, v in hb_alg.items()}
            t_all = sum(by[k]["total_ms"] for k in hf) / max(1, by[main_k]["calls"])
            t_main = by[main_k]["total_ms"] / by[main_k]["calls"]
            hb, hsrc = pmc_traffic(main_k)
            roof["hfre"] = dict(bound="hbm", kernel=" + ".join(sorted(hf)), peak=HBM_PEAK_GBS, unit="GB/s",
                                algorithmic_bytes=hb_alg["footprint_union"], algorithmic_bytes_full_map_upper_bound=hb_alg["full_map_upper_bound"],
                                achieved=round(hb_alg["footprint_union"] / (t_all * 1e-3) / 1e9, 1),
                                frac=round(hb_alg["footprint_union"] / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                achieved_main_kernel_only=round(hb_alg["footprint_union"] / (t_main * 1e-3) / 1e9, 1),
                                achieved_on_upper_bound_bytes=round(hb_alg["full_map_upper_bound"] / (t_all * 1e-3) / 1e9, 1),
                                us_all_launches=round(t_all * 1e3, 2), us_main_kernel=round(t_main * 1e3, 2), launches=len(hf),
                                images_per_launch=ipl, us_per_image=round(t_all * 1e3 / ipl, 2),
                                traffic=hb, traffic_source=hsrc)

    if rank == 0:
        n_img = args.steps * world * B
        out = dict(metric="images/sec", value=n_img / el, unit="images/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=el / args.steps * 1e3, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype=("fp8-e4m3 linears, preset %s (%d weights; bf16 elsewhere)" % (args.fp8, n_fp8)) if args.fp8 else "bf16", data="synthetic",
                   region_tokens_per_sec=n_img * args.boxes / el,
                   config=dict(workload=f"{'BASELINE metric config (100 boxes/img, COCO-typical 640x480)' if (img_hw == (480, 640) and args.boxes == 100) else ('BASELINE configs[1]' if (img_hw == (480, 640) and args.boxes == 32) else ('BASELINE configs[4] geometry (high-res dual encoder, 300 proposals/image)' if (img_hw == (1344, 1344) and args.boxes == 300) else 'non-default geometry'))}: 1 image "
                                        f"{img_hw[1]}x{img_hw[0]} (S={case['grid'][0] * case['grid'][1]} patches) x {args.boxes} proposals "
                                        f"(CountBench / Pixmo UPN boxes" + (f", run as {len(case['prompts'])} prompts of <= 100 over the one image: the reference caps region features at 100 per prompt, "
                                        "mm_utils.py:600 — towers once per image, one LLM sequence per prompt" if "prompts" in case else
                                        (", ALL in one prompt: a labelled EXTENSION, the reference caps region features at 100 per prompt and cannot run this" if args.boxes > 100 else "")) + "), Qwen2.5-VL-3B + DaViT-L + SimpleFPN true shapes, prompt "
                                        f"{len(case['ids']) - 1 + case['grid'][0] * case['grid'][1] // 4} tokens after splice, prefill to the first greedy token" +
                                        (" of every prompt" if "prompts" in case else ""),
                               stages=Pipeline.stages,
                               launch=("eager" if args.eager else "hipGraph replay (1 graph per shape signature)") +
                                      f"; a step = ONE packed pass over {B} different images (varlen batched prefill: rows of all images in every GEMM)" +
                                      (f"; {R} passes in flight on {R} HIP streams (engine replicas share weights)" if R > 1 else "; one pass at a time"),
                               images_per_step=B, passes_in_flight=R, global_batch=B * world,
                               parallelism=f"dp{world} (images sharded, no data-path collective)" + (" [test: all ranks on one device]" if one_dev else "")),
                   end_to_end=e2e, driver_level=drv, driver_level_countbench=drv_count, scale=scale, hires=hires, one_image_at_a_time=single, one_pass_at_a_time=one_pass, dataset=dset, decode=dec, preprocess=prep, roofline=roof)
        if args.fp8:
            out["fp8_note"] = ("W8A8 e4m3 linears are an MI355X-side lever BASELINE configs[4] names; the reference has no fp8 path, so this mode's parity is "
                               "UNPINNED (deviation table against the bf16 engine: DESIGN.md section 10, tests/test_fp8_engine_gpu.py)")
        if roof is not None:
            # stage times (sum of kernel execution time per stage, eager pass) and the two region-token rates
            out["stage_kernel_ms"] = stage_ms
            t_reg = stage_ms.get("hfre_region_pool", 0.0) + stage_ms.get("mm_projector_aux", 0.0)
            t_enc = t_reg + stage_ms.get("davit_large", 0.0) + stage_ms.get("simple_fpn", 0.0)
            if t_reg > 0:
                out["region_tokens_per_sec_hfre_plus_connector"] = round(B * args.boxes / (t_reg * 1e-3), 1)
                out["region_tokens_per_sec_encode_regions"] = round(B * args.boxes / (t_enc * 1e-3), 1)
            out["stage_kernel_ms_note"] = f"kernel time per packed pass of {B} images"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(case, pipe, reps=args.cpu_reps, decode_tokens=args.cpu_decode_tokens)
        emit(out, args.json_out)
    if world > 1:
        torch.distributed.barrier()   # ranks leave together (rank 0 was still measuring decode / roofline)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
