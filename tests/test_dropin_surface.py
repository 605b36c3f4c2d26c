"""CPU tests of the drop-in host surface (§8b): vlm_fo1.constants /
task_templates / mm_utils behave exactly like the reference's modules (imported in place when
/root/reference is present), and the known-answer cases hold everywhere."""
import importlib.util
import os
import random
import sys
import types

import pytest
import torch
from PIL import Image

import vlm_fo1.constants as C
import vlm_fo1.mm_utils as MU
import vlm_fo1.task_templates as T

REF = "/root/reference/vlm_fo1"
HAVE_REF = os.path.isdir(REF)


from vlm_fo1_amd.fixtures.synthetic import ToyTokenizer  # noqa: E402,F401  (shared with bench.py's driver_level block)


def load_ref(name):
    saved = {k: sys.modules.get(k) for k in ("vlm_fo1", "vlm_fo1.constants")}
    try:
        pkg = types.ModuleType("vlm_fo1"); pkg.__path__ = [REF]
        sys.modules["vlm_fo1"] = pkg
        spec = importlib.util.spec_from_file_location("vlm_fo1.constants", os.path.join(REF, "constants.py"))
        rc = importlib.util.module_from_spec(spec); spec.loader.exec_module(rc)
        sys.modules["vlm_fo1.constants"] = rc
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        return m
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
def test_constants_and_templates_equal_reference():
    rc, rt = load_ref("constants"), load_ref("task_templates")
    for n in [n for n in dir(rc) if n.isupper()]:
        assert getattr(C, n) == getattr(rc, n), n
    for n in [n for n in dir(rt) if n.endswith("template")]:
        assert getattr(T, n) == getattr(rt, n), n


def test_output_parsing_known_answers():
    s = "<ground>orange</ground><objects><region0><region2><region3></objects> and <ground> apple </ground><objects><region1></objects>" \
        "<ground>orange</ground><objects><region5><region2></objects>"
    got = MU.extract_predictions_to_indexes(s)
    assert got == {"orange": {0, 2, 3, 5}, "apple": {1}}
    boxes = [[i, i, i + 1, i + 1] for i in range(7)]
    gb = MU.extract_predictions_to_bboxes(s, boxes)
    assert sorted(map(tuple, gb["orange"])) == [(0, 0, 1, 1), (2, 2, 3, 3), (3, 3, 4, 4), (5, 5, 6, 6)] and gb["apple"] == [[1, 1, 2, 2]]
    assert MU.extract_predictions_to_indexes("no markup") == {}


def test_adjust_bbox_and_resize_known_answers():
    assert MU.adjust_bbox([[-5, 10, 700, 300]], 200, 400, 100, 200) == [[0.0, 5.0, 200.0, 100.0]]
    big = Image.new("RGB", (4096, 1024))
    imgs, boxes = MU.resize_shortest_edge_images_and_bboxes([big], [[0, 0, 4096, 1024]], max_size=2048)
    assert imgs[0].size == (2048, 512) and boxes == [[0.0, 0.0, 2048.0, 512.0]]
    small = Image.new("RGB", (500, 399))
    imgs, boxes = MU.resize_shortest_edge_images_and_bboxes([small], [[1, 2, 3, 4]], max_size=2048)
    assert imgs[0] is small and boxes == [[1.0, 2.0, 3.0, 4.0]]
    with pytest.raises(ValueError):
        MU.resize_shortest_edge_images_and_bboxes([], [], max_size=2048)


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
@pytest.mark.parametrize("bos", [None, 1])
def test_mm_utils_equivalent_to_reference(bos, tmp_path, capsys):
    ref = load_ref("mm_utils")
    tok = ToyTokenizer(bos)
    prompts = ["a <image> b <regionfeat> c <regionfeat>\n d", "<image>", "x y z", "<image><regionfeat><regionfeat> q <image> w <regionfeat>",
               "p <image_0> q <image_1> r"]
    for p in prompts:
        assert MU.tokenizer_image_token(p, tok) == ref.tokenizer_image_token(p, tok), p
        assert MU.tokenizer_image_region_token(p, tok) == ref.tokenizer_image_region_token(p, tok), p
        assert torch.equal(MU.tokenizer_image_token(p, tok, return_tensors="pt"), ref.tokenizer_image_token(p, tok, return_tensors="pt"))
    rnd = random.Random(0)
    boxes = [[rnd.uniform(-20, 700), rnd.uniform(-20, 500), rnd.uniform(0, 800), rnd.uniform(0, 600)] for _ in range(50)]
    assert MU.adjust_bbox([list(b) for b in boxes], 480, 640, 399, 500) == ref.adjust_bbox([list(b) for b in boxes], 480, 640, 399, 500)
    for size in [(500, 399), (4000, 3000), (20, 3000), (2049, 2049)]:
        im = Image.new("RGB", size)
        a = MU.resize_shortest_edge_images_and_bboxes([im], [list(b) for b in boxes], max_size=2048)
        b = ref.resize_shortest_edge_images_and_bboxes([im], [list(b) for b in boxes], max_size=2048)
        assert a[0][0].size == b[0][0].size and a[1] == b[1]
    texts = ["<ground>a b</ground><objects><region12><region3></objects>", "junk<ground>x</ground><objects></objects>",
             "<ground>c</ground><objects><region1></objects><ground>c</ground><objects><region9><region1></objects>"]
    bl = [[i, 0, i, 1] for i in range(20)]
    for t in texts:
        assert MU.extract_predictions_to_indexes(t) == ref.extract_predictions_to_indexes(t)
        assert MU.extract_predictions_to_bboxes(t, bl) == ref.extract_predictions_to_bboxes(t, bl)
    # make_message_context on the demo message layout (vision markers as prepare_inputs patches them)
    for mod in (MU, ref):
        mod.DEFAULT_IM_START_TOKEN, mod.DEFAULT_IM_END_TOKEN = "<|vision_start|>", "<|vision_end|>"
    for msg in [
        {"role": "user", "content": [{"type": "image_url", "image_url": {"url": "demo.jpg"}}, {"type": "text", "text": "Please detect orange"}],
         "bbox_list": [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]]},
        {"role": "user", "content": [{"type": "image_url", "image_url": {"url": "demo.jpg"}}, {"type": "text", "text": "describe"}]},
        {"role": "user", "content": "plain question"},
        {"role": "system", "content": "You are terse."},
    ]:
        assert MU.make_message_context(tok, msg) == ref.make_message_context(tok, msg)
    # stop criterion
    ids = torch.tensor([[5, 6, 7, 8]])
    a = MU.KeywordsStoppingCriteria(["<|im_end|>"], tok, ids)
    b = ref.KeywordsStoppingCriteria(["<|im_end|>"], tok, ids)
    kw = tok("<|im_end|>").input_ids[-1]
    for tail in ([9, 9], [9, kw]):
        out = torch.tensor([[5, 6, 7, 8] + tail])
        assert a(out, None) == b(out, None)


def test_prepare_inputs_keys_and_shapes(tmp_path):
    """prepare_inputs returns exactly the reference's kwargs keys (mm_utils.py:640-654) with engine-ready tensors."""
    from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor, Qwen2VLPatchProcessor
    img = Image.effect_noise((500, 399), 64).convert("RGB")
    path = str(tmp_path / "demo.jpg")
    img.save(path)
    tok = ToyTokenizer()
    model = types.SimpleNamespace(config=types.SimpleNamespace(mm_use_region_index_token=True))
    boxes = [[161.0, 11.0, 292.0, 127.0], [268.0, 61.0, 428.0, 226.0]]
    msgs = [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": path}}, {"type": "text", "text": T.OD_template.format("orange")}],
             "bbox_list": boxes}]
    kw = MU.prepare_inputs("VLM-FO1_Qwen2.5-VL-3B-v01", model, (Qwen2VLPatchProcessor(), CLIPStyleAuxProcessor(resize_mode="dynamic")),
                           tok, msgs, device="cpu", max_tokens=64)
    assert sorted(kw) == sorted(["inputs", "images", "images_aux", "image_grid_thws", "bbox_list", "do_sample", "temperature",
                                 "max_new_tokens", "streamer", "top_p", "use_cache", "stopping_criteria", "pad_token_id"])
    ids = kw["inputs"][0].tolist()
    assert ids.count(C.IMAGE_TOKEN_INDEX) == 1 and ids.count(C.DEFAULT_REGION_INDEX) == 2
    assert kw["images"][0].shape == (28 * 36, 1176) and kw["image_grid_thws"][0].tolist() == [[1, 28, 36]]
    assert kw["images_aux"][0].shape == (3, 399, 500) and kw["bbox_list"][0].shape == (2, 4)
    assert kw["do_sample"] is False and kw["max_new_tokens"] == 64 and kw["use_cache"] is True


def _noise_image(w, h, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))


@pytest.mark.parametrize("w,h,max_pixels", [(500, 399, 2048 * 2048), (640, 480, 2048 * 2048), (30, 40, 2048 * 2048),
                                            (700, 530, 512 * 512)])   # last: the area > max_pixels branch, scaled down to stay fast
def test_primary_processor_matches_hf_pil_processor(w, h, max_pixels):
    """Our smart-resize + patchify vs the installed HF `Qwen2VLImageProcessorPil`."""
    try:
        from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    except Exception:
        pytest.skip("HF PIL processor not importable")
    from vlm_fo1.model.image_processing import Qwen2VLPatchProcessor
    img = _noise_image(w, h, w + h)
    hf = Qwen2VLImageProcessorPil(size={"shortest_edge": 56 * 56, "longest_edge": max_pixels})
    ref = hf(images=img, return_tensors="pt")
    got = Qwen2VLPatchProcessor(max_pixels=max_pixels).preprocess(img, videos=None, return_tensors="pt")
    assert torch.equal(got["image_grid_thw"], ref["image_grid_thw"])
    torch.testing.assert_close(got["pixel_values"], ref["pixel_values"].float(), rtol=0, atol=2e-6)


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
@pytest.mark.parametrize("mode", ["dynamic", "squash", "dynamic_square"])
def test_aux_processor_matches_reference_clip_processor(mode):
    spec = importlib.util.spec_from_file_location("ref_clip_ip", os.path.join(REF, "model/multimodal_encoder/davit/image_processing_clip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    cfg = dict(do_resize=mode != "dynamic", size={"height": 768, "width": 768}, resample=3, do_center_crop=False, do_rescale=True,
               do_normalize=True, image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], do_convert_rgb=True,
               resize_mode=mode)
    ref_ip = m.CLIPImageProcessor(**cfg)
    from vlm_fo1.model.image_processing import CLIPStyleAuxProcessor
    # (dynamic_square, image_processing_clip.py:190-204: the candidate square closest in AREA — 500 x 399 -> 448, 700 x 530 -> 640, and
    # 448 x 577 = 258 496 sits between 512^2 and 448^2 ... every size goes through the reference's own selection loop)
    for w, h, seed in ((500, 399, 3), (700, 530, 4), (448, 577, 5)) if mode == "dynamic_square" else ((500, 399, 3),):
        img = _noise_image(w, h, seed)
        ref = ref_ip.preprocess(img, return_tensors="pt")["pixel_values"][0]
        got = CLIPStyleAuxProcessor(resize_mode=mode).preprocess(img, return_tensors="pt")["pixel_values"][0]
        assert got.shape == ref.shape, (mode, w, h, got.shape, ref.shape)
        torch.testing.assert_close(got, ref, rtol=0, atol=2e-6)


def test_model_max_length_overflow_fails_like_the_reference_splice():
    """omchat_qwen2_5_vl.py:374-378 cuts the spliced embeddings to tokenizer_model_max_length but not new_input_ids, so the
    right-padding fill at :411 raises torch's size-mismatch RuntimeError: an over-long prompt never reaches the LLM.  The drop-in
    raises the same exception type with the same two sizes; prompts within the limit pass; left padding is refused explicitly."""
    from vlm_fo1.model.fo1_model import FO1ForCausalLM, FO1HFConfig
    from vlm_fo1_amd.llm import DEFAULT_REGION_INDEX, IMAGE_TOKEN_INDEX
    m = FO1ForCausalLM.__new__(FO1ForCausalLM)
    m.config = FO1HFConfig({"tokenizer_model_max_length": 100})
    ids = [1, 2, IMAGE_TOKEN_INDEX, 3] + [DEFAULT_REGION_INDEX, 7] * 10          # 24 ids, one image sentinel
    m._check_model_max_length(ids, 77)                                            # 24 + 76 = 100 rows: fits exactly
    with pytest.raises(RuntimeError, match=r"\(100\) must match the existing size \(101\)"):
        m._check_model_max_length(ids, 78)
    # the reference's statement really fails that way
    import torch
    with pytest.raises(RuntimeError, match=r"\(100\) must match the existing size \(101\)"):
        torch.zeros(1, 200, dtype=torch.long)[0, :100] = torch.zeros(101, dtype=torch.long)
    m.config = FO1HFConfig({"tokenizer_model_max_length": 100, "tokenizer_padding_side": "left"})
    with pytest.raises(NotImplementedError):
        m._check_model_max_length(ids, 78)
    m.config = FO1HFConfig({})
    m._check_model_max_length(ids, 10 ** 6)                                       # no limit configured: nothing to check


def test_chunk_tokenisation_cache_is_transparent():
    """The per-tokenizer chunk cache returns what the tokenizer returns, calls it once per distinct chunk, and hands
    out fresh lists (callers extend them)."""
    from vlm_fo1 import mm_utils

    class Tok:
        bos_token_id = None

        def __init__(self):
            self.calls = 0

        def __call__(self, text):
            self.calls += 1
            return type("E", (), {"input_ids": [ord(c) for c in text]})()

    tok = Tok()
    prompt = "a<image>b" + "".join(f"<region{i}><regionfeat>" for i in range(20)) + "c"
    first = mm_utils.tokenizer_image_region_token(prompt, tok)
    n = tok.calls
    again = mm_utils.tokenizer_image_region_token(prompt, tok)
    assert again == first and tok.calls == n, "second call must be served from the cache"
    ref = mm_utils.tokenizer_image_region_token(prompt, Tok())       # a fresh tokenizer object: nothing cached
    assert ref == first
    first.append(-1)
    assert mm_utils.tokenizer_image_region_token(prompt, tok)[-1] != -1


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
@pytest.mark.parametrize("script", ["inference.py", "scripts/inference_with_upn.py", "scripts/run_upn.py", "evaluation/eval_coco.py",
                                    "evaluation/eval_countbench.py"])
def test_every_name_the_reference_drivers_import_exists_in_the_drop_in(script):
    """The reference's driver scripts run unmodified against this repo's `vlm_fo1` / `detect_tools` packages: every
    `from vlm_fo1... import name` / `from detect_tools... import name` they contain resolves here (callables stay callables)."""
    import ast
    import importlib
    src = open(os.path.join("/root/reference", script)).read()
    checked = 0
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in ("vlm_fo1", "detect_tools"):
            mod = importlib.import_module(node.module)
            assert os.path.abspath(mod.__file__).startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), mod.__file__
            for a in node.names:
                if a.name == "*":
                    continue
                assert hasattr(mod, a.name), f"{script}: {node.module}.{a.name} is missing from the drop-in"
                checked += 1
    assert checked > 0
