"""COCO-val proposal classification with the MI355X engine — same CLI, inputs and output file as the
reference's evaluation/eval_coco.py:12-100 (per image: prompt with the UPN proposals -> generate -> regex ->
COCO-json records with the ORIGINAL proposal box and its UPN score), plus sharding over the GPUs of one node:

    python evaluation/eval_coco.py --model_id resources/VLM-FO1_Qwen2.5-VL-3B-v01 ...            # 1 GPU
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 evaluation/eval_coco.py ...               # 8 GPUs

Images are dealt to ranks by cost; the only collective is the final gather of generated token ids
(vlm_fo1_amd/sharded_eval.py); rank 0 writes a dump byte-identical to the 1-GPU run."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from vlm_fo1.mm_utils import extract_predictions_to_indexes, prepare_inputs  # noqa: E402
from vlm_fo1.model.builder import load_pretrained_model  # noqa: E402
from vlm_fo1_amd import sharded_eval as SE  # noqa: E402


def records_from_answer(ans, data, cat_ids):
    """One image's COCO records (reference eval_coco.py:68-85)."""
    out = []
    for label, idxs in extract_predictions_to_indexes(ans).items():
        for b in idxs:
            box, score = data["bbox_list"][b], data["score_list"][b]
            if label in cat_ids:
                out.append({"image_id": data["id"], "category_id": cat_ids[label],
                            "bbox": [box[0], box[1], box[2] - box[0], box[3] - box[1]], "score": score})
    return out


def eval_coco(model_id, eval_data_path, original_data_path, img_folder, out_dir=None, device="cuda:0"):
    # host tensors on this path are a few hundred KB at most: one intra-op thread.  With torch's default (one per core, 256 here) every
    # small torch.cat / clone of the prefetch and worker threads opens an OpenMP region on the same oversubscribed pool — measured 10x
    # slower host planning and 50 ms per prepare_inputs (profiles/r04_driver_level_host_profile_*.log)
    torch.set_num_threads(int(os.environ.get("FO1_HOST_TORCH_THREADS", "1")))
    sys.setswitchinterval(float(os.environ.get("FO1_SWITCH_INTERVAL", "0.0005")))   # a dozen short-burst host threads: 5 ms GIL hand-overs starve the launch threads
    rank, world, local = SE.init_distributed()
    if world > 1:
        device = f"cuda:{local}"
    print(f"Evaluating {model_id} on {eval_data_path}... (rank {rank}/{world})")
    tokenizer, model, image_processors = load_pretrained_model(model_id, device=device)
    with open(eval_data_path) as f:
        data_list = [json.loads(line) for line in f]
    cat_ids = {c["name"]: c["id"] for c in json.load(open(original_data_path))["categories"]}

    batch = max(1, int(os.environ.get("FO1_BATCH", "32")))     # images per packed pass (prefill + batched decode); 1 = the reference's loop

    def inputs_of(i):
        """Host side of one item (a1): PIL decode / resize, tokenisation, uploads.  Runs on the prefetch threads, ahead of the GPU."""
        d = data_list[i]
        messages = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": os.path.join(img_folder, d["image"])}},
                                                 {"type": "text", "text": d["conversations"][0]["value"]}],
                     "bbox_list": d["bbox_list"]}]
        kw = prepare_inputs(model_id, model, image_processors, tokenizer, messages, device=device, max_tokens=int(os.environ.get("FO1_MAX_NEW_TOKENS", "4096")), top_p=0.05,
                            temperature=0.0, do_sample=False)
        kw["streamer"] = None
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()      # uploads / device preprocessing done before another stream consumes them
        return kw

    def make_generate(m, stream):
        def generate(i, kw):
            with torch.cuda.stream(stream):
                out = m.generate(**kw)
                return out[0, kw["inputs"].shape[1]:].tolist()

        def generate_group(idxs, kws):
            with torch.cuda.stream(stream):
                if hasattr(m, "generate_many_async"):       # prefill now, decode in the shared pool; the worker goes on with its next group
                    pend = m.generate_many_async(kws)
                    return SE.Deferred(lambda: [o[0, kw["inputs"].shape[1]:].tolist() for o, kw in zip(pend.result(), kws)])
                outs = m.generate_many(kws)
                return [o[0, kw["inputs"].shape[1]:].tolist() for o, kw in zip(outs, kws)]     # (the reference's slice, inference.py:47-48)
        return generate_group if batch > 1 else generate

    generate = SE.request_workers(model, make_generate)       # $FO1_INFLIGHT passes in flight per GPU (default 2: each keeps up to three groups in the pool), one decode pool ($FO1_DECODE_POOL)
    # cost = f(pixels, N): the image header gives the size without decoding; a missing file falls back to the box count
    costs = [SE.item_cost(*SE.image_size(os.path.join(img_folder, d["image"])), len(d["bbox_list"])) for d in data_list]
    merged = SE.run_sharded(len(data_list), costs, generate, device=device if world > 1 else "cpu", batch=batch, prepare=inputs_of,
                            prefetch_depth=max(16, 3 * batch), prefetch_threads=int(os.environ.get("FO1_PREFETCH_THREADS", "4")))
    if rank != 0:
        return
    res = []
    for i, toks in merged:
        if toks is None:
            print(f"Error: {data_list[i]['id']}")
            continue
        ans = tokenizer.decode(toks).strip()
        res.extend(records_from_answer(ans, data_list[i], cat_ids))
    output_path = os.path.join(out_dir, model_id.split("/")[-1])
    os.makedirs(output_path, exist_ok=True)
    out_file = f"{output_path}/{eval_data_path.split('/')[-1].replace('.jsonl', '')}_predictions.json"
    json.dump(res, open(out_file, "w"))
    print(f"predictions saved to: {out_file}")


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--model_id", type=str, default="resources/VLM-FO1_Qwen2.5-VL-3B-v01")
    p.add_argument("--eval_data_path", type=str, default="evaluation/processed_data/cocoVal2017_with_upn_score_0.3_0.8.jsonl")
    p.add_argument("--original_data_path", type=str, default="evaluation/processed_data/instances_val2017.json")
    p.add_argument("--img_folder", type=str, default="data/coco/val2017")
    p.add_argument("--out_dir", type=str, default="./evaluation")
    p.add_argument("--device", type=str, default="cuda:0")
    a = p.parse_args()
    eval_coco(a.model_id, a.eval_data_path, a.original_data_path, a.img_folder, a.out_dir, a.device)
