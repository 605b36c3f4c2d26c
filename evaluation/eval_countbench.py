"""CountBench / Pixmo-Count accuracy with the MI355X engine — same CLI and scoring as the reference's
evaluation/eval_countbench.py:14-76 (first integer of the answer after stripping <regionN> tags), sharded over
the GPUs of one node like eval_coco.py (torchrun --nproc-per-node N).  Variable box counts (2..100 per image)
are balanced across ranks by cost."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from vlm_fo1.mm_utils import prepare_inputs  # noqa: E402
from vlm_fo1.model.builder import load_pretrained_model  # noqa: E402
from vlm_fo1_amd import sharded_eval as SE  # noqa: E402


def count_from_answer(outputs: str) -> int:
    """reference eval_countbench.py:48-53"""
    ans = re.sub(r"<region\d+>", "", outputs)
    numbers = re.findall(r"(?<!region)\d+", ans)
    return int(numbers[0]) if numbers else 0


def eval_countbench(data_path, image_path, model_id, device):
    # host tensors on this path are a few hundred KB at most: one intra-op thread.  With torch's default (one per core, 256 here) every
    # small torch.cat / clone of the prefetch and worker threads opens an OpenMP region on the same oversubscribed pool — measured 10x
    # slower host planning and 50 ms per prepare_inputs (profiles/r04_driver_level_host_profile_*.log)
    torch.set_num_threads(int(os.environ.get("FO1_HOST_TORCH_THREADS", "1")))
    sys.setswitchinterval(float(os.environ.get("FO1_SWITCH_INTERVAL", "0.0005")))   # a dozen short-burst host threads: 5 ms GIL hand-overs starve the launch threads
    rank, world, local = SE.init_distributed()
    if world > 1:
        device = f"cuda:{local}"
    tokenizer, model, image_processors = load_pretrained_model(model_id, device=device)
    with open(data_path) as f:
        data = json.load(f)

    batch = max(1, int(os.environ.get("FO1_BATCH", "32")))     # images per packed pass (prefill + batched decode); 1 = the reference's loop

    def inputs_of(i):
        """Host side of one item (a1): PIL decode / resize, tokenisation, uploads.  Runs on the prefetch threads, ahead of the GPU."""
        item = data[i]
        messages = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": os.path.join(image_path, item["image"])}},
                                                 {"type": "text", "text": item["question"]}], "bbox_list": item["bboxes"]}]
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
    # cost = f(pixels, N): the image header gives the size without decoding; a missing file falls back to the box extent
    costs = [SE.item_cost(*SE.image_size(os.path.join(image_path, item["image"]), item["bboxes"]), len(item["bboxes"])) for item in data]
    merged = SE.run_sharded(len(data), costs, generate, device=device if world > 1 else "cpu", batch=batch, prepare=inputs_of,
                            prefetch_depth=max(16, 3 * batch), prefetch_threads=int(os.environ.get("FO1_PREFETCH_THREADS", "4")))
    if rank != 0:
        return None
    correct = total = 0
    for i, toks in merged:
        gt = data[i]["answer"]
        outputs = tokenizer.decode(toks).strip() if toks is not None else ""
        pred = count_from_answer(outputs)
        total += 1
        correct += int(pred == gt)
        if gt != pred:
            print(f"gt is {gt}, but pred is {outputs}")
    accuracy = correct / total if total > 0 else 0
    print(f"Accuracy: {accuracy:.4f}")
    return accuracy


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--data_path", type=str, default="evaluation/processed_data/countbench_with_upn_score_0.3_0.8.json")
    p.add_argument("--image_path", type=str, default="data/CountBenchQA/images")
    p.add_argument("--model_id", type=str, default="resources/VLM-FO1_Qwen2.5-VL-3B-v01")
    p.add_argument("--device", type=str, default="cuda:0")
    a = p.parse_args()
    eval_countbench(a.data_path, a.image_path, a.model_id, a.device)
