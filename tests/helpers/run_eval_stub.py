"""Test driver: runs evaluation/eval_coco.py or eval_countbench.py END TO END (data loading, sharding over the ranks, the one gather,
decode / regex / dump on rank 0) with a CPU stub in place of the engine-backed model — the eval drivers' own code is what runs;
only `load_pretrained_model` / `prepare_inputs` are replaced (no GPU in the CPU test tier).  Launched by tests/test_eval_drivers_cpu.py
under torch.distributed.run with 1 and 2 ranks (gloo).

The stub "model" answers deterministically from the item's boxes: it grounds the label named in the question to every third
region, so the parse -> COCO-record path sees real `<ground>..</ground><objects><regionK>..</objects>` markup; item 3 raises (per-item
error record path); ids are characters + 100 (a toy tokenizer that decodes them back)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "evaluation"))

import torch  # noqa: E402


class Tok:
    def decode(self, ids, **kw):
        return "".join(chr(int(i) - 100) for i in ids)


class StubModel:
    device = torch.device("cpu")

    def __init__(self):
        self.calls = 0

    def _answer(self, kw):
        meta = kw["_meta"]
        if meta["fail"]:
            raise RuntimeError("synthetic failure")
        regions = "".join(f"<region{k}>" for k in range(0, meta["n_boxes"], 3))
        text = f"<ground>{meta['label']}</ground><objects>{regions}</objects> {meta['n_boxes']}"
        return torch.tensor([[ord(c) + 100 for c in text]], dtype=torch.long)

    def generate(self, **kw):
        self.calls += 1
        return torch.cat([kw["inputs"], self._answer(kw)], dim=1)

    def generate_many(self, kws):
        self.calls += 1
        return [torch.cat([kw["inputs"], self._answer(kw)], dim=1) for kw in kws]


def stub_prepare_inputs(model_id, model, procs, tokenizer, messages, **unused):
    content = messages[0]["content"]
    text = [c["text"] for c in content if c["type"] == "text"][0]
    url = [c["image_url"]["url"] for c in content if c["type"] == "image_url"][0]
    label = text.split("LABEL=")[1].split()[0] if "LABEL=" in text else "person"
    return dict(inputs=torch.arange(5).view(1, -1), max_new_tokens=64,
                _meta=dict(n_boxes=len(messages[0]["bbox_list"]), label=label, fail=url.endswith("img3.jpg")))


def main():
    which, *rest = sys.argv[1:]
    import vlm_fo1_amd.sharded_eval as SE
    SE.request_workers = lambda model, make_generate, n=None: [make_generate(model, None)]   # no GPU: one worker, no stream
    if which == "coco":
        import eval_coco as E
        E.load_pretrained_model = lambda model_id, device="cpu": (Tok(), StubModel(), None)
        E.prepare_inputs = stub_prepare_inputs
        E.torch.cuda.stream = lambda s: __import__("contextlib").nullcontext()
        eval_path, orig_path, out_dir = rest
        E.eval_coco("stub/VLM-FO1_stub", eval_path, orig_path, "imgs", out_dir, device="cpu")
    else:
        import eval_countbench as E
        E.load_pretrained_model = lambda model_id, device="cpu": (Tok(), StubModel(), None)
        E.prepare_inputs = stub_prepare_inputs
        E.torch.cuda.stream = lambda s: __import__("contextlib").nullcontext()
        data_path, out_file = rest
        acc = E.eval_countbench(data_path, "imgs", "stub/VLM-FO1_stub", "cpu")
        if acc is not None:
            json.dump({"accuracy": acc}, open(out_file, "w"))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
