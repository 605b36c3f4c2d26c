"""Seeded full-depth workloads shared by the reference-golden generator (tests/golden/make_fulldepth_ref.py, build container) and
the GPU parity tests (tests/test_fulldepth_parity_gpu.py, GPU box).

Weights are generated ON THE CPU (torch's CPU generator is the one both machines share; the device stream differs) at the true
Qwen2.5-VL-3B + DaViT-L + SimpleFPN shapes and uploaded; a checksum per part detects RNG drift instead of silently comparing
different models.  The LM head is UNTIED and peaked: a seeded N(0, 0.02) matrix with HEAD_LIVE = 8 "live" rows scaled 12x — most
steps are decided among a handful of candidates by the hidden state's direction, the way a trained head decides, instead of being a
152k-way near-tie of an iid head whose argmax NO bf16 execution can pin (relative top-1 gap ~1 / (2 ln n) = 4 %, the size of the bf16
noise after 36 layers).  Untied, because a tied peaked embedding would feed large-norm rows back as inputs and make
every continuation repeat its own token.

Cases (BASELINE.json configs):
  metric  640x480 x 100 CountBench proposals      — the configuration `metric` is quoted on
  demo    500x399 x the 7 boxes of inference.py:16 — configs[0]
  hires   1344x1344 x 300 proposals as 3 x 100     — configs[4]; the reference caps features at 100 per prompt (mm_utils.py:600)"""
import torch

from hfre_cases import DEMO_BOXES, box_fixtures

HEAD_SEED, HEAD_SIGMA, HEAD_CLAMP, HEAD_LIVE, HEAD_LIVE_SCALE = 4321, 0.0, 1.0, 8, 12.0
K_DECODE = 16
CASES = {"metric": dict(img_hw=(480, 640), n_boxes=100, seed=77),
         "demo": dict(img_hw=(399, 500), n_boxes=7, seed=78),
         "hires": dict(img_hw=(1344, 1344), n_boxes=300, seed=79)}


def full_config():
    from vlm_fo1_amd.model import FO1Config
    return FO1Config()


def peaked_head(vocab: int, hidden: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(HEAD_SEED)
    w = torch.randn(vocab, hidden, generator=g) * 0.02
    s = torch.exp(HEAD_SIGMA * torch.randn(vocab, generator=g)).clamp(max=HEAD_CLAMP)
    live = torch.randperm(vocab, generator=g)[:HEAD_LIVE]
    s[live] = HEAD_LIVE_SCALE
    return (w * s[:, None]).to(torch.bfloat16)


def part_checksum(sd) -> int:
    s = 0
    for name in sorted(sd):
        s = (s * 1000003 + int(sd[name].contiguous().view(torch.int16).to(torch.int64).sum())) % (1 << 61)
    return s


def cpu_weights(cfg=None, seed: int = 0):
    """-> (weights dict of CPU bf16 tensors incl. llm['lm_head.weight'], {part: checksum})."""
    from vlm_fo1_amd.model import random_weights
    cfg = cfg or full_config()
    W = random_weights(cfg, "cpu", seed=seed)
    W["llm"]["lm_head.weight"] = peaked_head(cfg.llm.vocab_size, cfg.llm.hidden_size)
    return W, {k: part_checksum(v) for k, v in W.items()}


def build_case(name: str):
    """-> dict(pix [S,1176] bf16, aux [3,H,W] bf16, grid, img_hw, groups=[(ids, boxes [n,4] fp32)]) — one group per prompt (hires: 3)."""
    from vlm_fo1_amd.model import synthetic_prompt
    c = CASES[name]
    H, W = c["img_hw"]
    g = torch.Generator().manual_seed(c["seed"])
    gh, gw = round(H / 28) * 2, round(W / 28) * 2
    pix = torch.randn(gh * gw, 1176, generator=g).bfloat16()
    aux = torch.randn(3, H, W, generator=g).bfloat16()
    if name == "demo":
        boxes = torch.tensor(DEMO_BOXES, dtype=torch.float32)
    else:
        fx = box_fixtures()
        items = sorted(fx["countbench"] + fx["pixmo"], key=lambda x: -len(x["bboxes"]))      # the 100-box items first
        chunks = []
        for it in items:
            b = torch.tensor(it["bboxes"], dtype=torch.float32)[:100]
            ex, ey = it["extent"]
            chunks.append(b * torch.tensor([W / ex, H / ey, W / ex, H / ey]))   # the rescale adjust_bbox does (mm_utils.py:296-311)
        boxes = torch.cat(chunks)[:c["n_boxes"]]
        assert boxes.shape[0] == c["n_boxes"]
    groups = []
    for k in range(0, boxes.shape[0], 100):
        b = boxes[k:k + 100]
        groups.append((synthetic_prompt(b.shape[0], n_text=60, seed=c["seed"] + k, lead_seed=c["seed"]), b))     # the prompts of an image share its preamble
                                                                                                         # (prompt 0, the golden's, is unchanged)
    return dict(name=name, pix=pix, aux=aux, grid=(gh, gw), img_hw=(H, W), boxes=boxes, groups=groups)
