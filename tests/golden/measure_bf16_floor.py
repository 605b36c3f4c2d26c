"""Measures the REFERENCE's own bf16 noise floor at full depth on the metric's configuration (640x480 x 100 boxes):
the reference's modules (vendored Qwen2.5-VL ViT + custom_forward/VisionFeaturesGather, DaViT-L, SimpleFP imported in place from
/root/reference; HF Qwen2_5_VLTextModel for the LLM — the vendored LLM half does not construct under this transformers; HFRE through
oracle/hfre_oracle.py, which is pinned to the reference's HFREModule and — like it — computes in fp32 on `.float()` maps) are run
TWICE on the CPU with the same seeded weights and inputs: once the way the reference executes (`model.to(bfloat16)`,
builder.py:140-141) and once in fp32.  The per-stage deviation bf16-vs-fp32 is what ANY bf16 implementation of this path carries
against an fp32 evaluation; tests/test_fulldepth_parity_gpu.py uses it as the written-down floor where it is looser than the
SURVEY §7 starting tolerances.

    python tests/golden/measure_bf16_floor.py        # ~10-20 min on 8 cores; writes tests/golden/bf16_floor.json

Weights: vlm_fo1_amd.model.random_weights(FO1Config(), "cpu", seed=0) — the distribution the GPU tests and bench.py use (the
device RNG stream differs from the CPU one, so the values differ; the floor is a property of the distribution and depth)."""
import json
import os
import sys
import time
from unittest import mock

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from oracle import davit_oracle as DO, hfre_oracle as HO, llm_oracle as LO, reference_loader as R  # noqa: E402
from vlm_fo1_amd.model import FO1Config, random_weights  # noqa: E402

noop = lambda t, *a, **k: t


def metrics(got, ref):
    got, ref = got.float().reshape(-1, got.shape[-1]), ref.float().reshape(-1, ref.shape[-1])
    cos = F.cosine_similarity(got, ref, dim=-1)
    return dict(min_cos=float(cos.min()), mean_cos=float(cos.mean()), rel=float((got - ref).abs().max() / ref.abs().max()),
                rms_rel=float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))


def build_modules(cfg, W, with_llm=True):
    qwen, enc = R.vendored_qwen(), R.vendored_vit_encoder()
    vc = qwen.Qwen2_5_VLVisionConfig(depth=cfg.vit.depth, hidden_size=1280, hidden_act="silu", intermediate_size=3420, num_heads=16,
                                     in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
                                     out_hidden_size=2048, fullatt_block_indexes=list(cfg.vit.fullatt_block_indexes))
    vc._attn_implementation = "sdpa"
    vit = qwen.Qwen2_5_VisionTransformerPretrainedModel._from_config(vc, attn_implementation="sdpa").eval().float()
    vit.load_state_dict({k: v.float() for k, v in W["vit"].items()}, strict=True)
    dv, cfgs = R.vendored_davit()
    dc = cfgs.model_configs["davit-large"]
    with mock.patch.object(dv, "trunc_normal_", noop), mock.patch("torch.nn.init.normal_", noop), \
            mock.patch("torch.nn.init.kaiming_uniform_", noop), mock.patch("torch.nn.init.uniform_", noop), \
            mock.patch("torch.nn.init.constant_", noop):
        davit = dv.DaViT(depths=dc["depths"], embed_dims=dc["dim_embed"], num_heads=dc["num_heads"], num_groups=dc["num_groups"],
                         patch_size=dc["patch_size"], patch_stride=dc["patch_stride"], patch_padding=dc["patch_padding"],
                         patch_prenorm=dc["patch_prenorm"], window_size=dc["window_size"]).eval()
    davit.load_state_dict({k: v.float() for k, v in W["davit"].items()}, strict=True)
    _, SimpleFP, _ = HO.load_reference_hfre()
    fpn = SimpleFP(out_channels=512, norm="LN", square_pad=0, dim=1280, stride=14).eval()
    fpn.load_state_dict({k: v.float() for k, v in W["fpn"].items()}, strict=True)
    if not with_llm:
        return dict(vit=vit, enc=enc, davit=davit, fpn=fpn, llm=None, proj=_projectors(W))
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig
    lc = Qwen2_5_VLTextConfig(vocab_size=cfg.llm.vocab_size, hidden_size=2048, intermediate_size=11008, num_hidden_layers=cfg.llm.num_layers,
                              num_attention_heads=16, num_key_value_heads=2, max_position_embeddings=4096, rms_norm_eps=1e-6,
                              rope_theta=1e6, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                              rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, tie_word_embeddings=True)
    with mock.patch("torch.nn.init.normal_", noop), mock.patch("torch.nn.init.kaiming_uniform_", noop), mock.patch("torch.nn.init.uniform_", noop):
        llm = M.Qwen2_5_VLTextModel(lc).eval()
    llm.load_state_dict({k: v.float() for k, v in W["llm"].items()}, strict=True)
    return dict(vit=vit, enc=enc, davit=davit, fpn=fpn, llm=llm, proj=_projectors(W))


def _projectors(W):
    """mlp2x_gelu as build_vision_projector(_aux) builds it (multimodal_projector/builder.py:64-71,103-110)."""
    proj = {}
    for name in ("mm_projector", "mm_projector_aux"):
        w0, w2 = W["proj"][name + ".0.weight"], W["proj"][name + ".2.weight"]
        m = torch.nn.Sequential(torch.nn.Linear(w0.shape[1], w0.shape[0]), torch.nn.GELU(), torch.nn.Linear(w2.shape[1], w2.shape[0])).eval()
        m.load_state_dict({"0.weight": w0.float(), "0.bias": W["proj"][name + ".0.bias"].float(),
                           "2.weight": w2.float(), "2.bias": W["proj"][name + ".2.bias"].float()})
        proj[name] = m
    return proj


def one_pass(mods, case, cfg, dtype, embed_w, log):
    """The reference's data flow with every module in `dtype`; HFRE in fp32 on .float() maps like the reference."""
    gh, gw = case["grid"]
    H, Wd = case["img_hw"]
    out = {}
    t0 = time.perf_counter()
    vit, enc = mods["vit"].to(dtype), mods["enc"]
    gather = enc.VisionFeaturesGather()
    vit.vision_features_gather = gather
    with torch.no_grad():
        tokens = enc.custom_forward(vit, case["pix"].to(dtype), torch.tensor([[1, gh, gw]]))
        maps = gather.extract_multi_level_features()[0]
        out["vit_tokens"] = tokens.float()
        out["vit_map"] = maps[-1][0].permute(1, 2, 0).reshape(gh * gw, -1).float()
        log(f"{dtype}: vit {time.perf_counter() - t0:.1f}s")
        t0 = time.perf_counter()
        img_tok = mods["proj"]["mm_projector"].to(dtype)(tokens)
        out["image_tokens"] = img_tok.float()
        fpn = mods["fpn"].to(dtype)(maps[-1])
        for i, m in enumerate(fpn):
            out[f"fpn_level{i}"] = m[0].permute(1, 2, 0).reshape(-1, m.shape[1]).float()
        log(f"{dtype}: fpn {time.perf_counter() - t0:.1f}s")
        t0 = time.perf_counter()
        aux = mods["davit"].to(dtype).forward_features(case["aux"].to(dtype).unsqueeze(0))["image_features"]
        for i, m in enumerate(aux):
            out[f"davit_stage{i}"] = m[0].permute(1, 2, 0).reshape(-1, m.shape[1]).float()
        log(f"{dtype}: davit {time.perf_counter() - t0:.1f}s")
        sw, sh = gw * 14 / Wd, gh * 14 / H
        boxes = case["boxes"]
        feat = HO.hfre_oracle([m for m in aux], boxes, [m for m in fpn], boxes * torch.tensor([sw, sh, sw, sh]), region_dim=5888,
                              grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
        out["hfre_features"] = feat.float()
        reg = mods["proj"]["mm_projector_aux"].to(dtype)(feat.to(dtype))          # omchat_qwen2_5_vl.py:106-107
        out["region_tokens"] = reg.float()
        emb, nb, na = LO.splice(torch.tensor(case["ids"]), embed_w.to(dtype), img_tok, reg)
        pos, _ = LO.rope_index(nb, (gh // 2, gw // 2), na)
        out["embeds"] = emb.float()
        t0 = time.perf_counter()
        llm = mods["llm"].to(dtype)
        keep = {}
        h = llm.layers[-1].register_forward_hook(lambda mod, a, o: keep.__setitem__("h", (o[0] if isinstance(o, tuple) else o)))
        final = llm(inputs_embeds=emb.to(dtype)[None], position_ids=pos[:, None, :]).last_hidden_state[0]
        h.remove()
        out["llm_hidden"] = keep["h"].reshape(-1, 2048).float()
        out["llm_final_last_row"] = final[-1:].float()
        out["logits"] = (final[-1:] @ embed_w.to(dtype).t()).float()
        log(f"{dtype}: llm {time.perf_counter() - t0:.1f}s")
        out["pos"] = pos
    return out


def llm_isolated(mods, emb, pos, embed_w, dtype):
    with torch.no_grad():
        llm = mods["llm"].to(dtype)
        keep = {}
        h = llm.layers[-1].register_forward_hook(lambda mod, a, o: keep.__setitem__("h", (o[0] if isinstance(o, tuple) else o)))
        final = llm(inputs_embeds=emb.to(dtype)[None], position_ids=pos[:, None, :]).last_hidden_state[0]
        h.remove()
        return keep["h"].reshape(-1, 2048).float(), final[-1:].float(), (final[-1:] @ embed_w.to(dtype).t()).float()


def main():
    assert R.available(), "/root/reference is needed"
    torch.set_num_threads(int(os.environ.get("FLOOR_THREADS", "8")))
    cfg = FO1Config()
    if os.environ.get("FLOOR_SMALL"):   # quick functional check of this script
        cfg.vit.depth, cfg.vit.fullatt_block_indexes, cfg.llm.num_layers, cfg.llm.vocab_size = 2, (1,), 2, 4096
    W = random_weights(cfg, "cpu", seed=0)
    case = bench.build_workload(None, n_boxes=100, seed=77)
    if os.environ.get("FLOOR_SMALL"):
        from vlm_fo1_amd.model import synthetic_prompt
        case["ids"] = synthetic_prompt(100, n_text=60, vocab=4096, seed=77)
    log = lambda s: print(s, flush=True)
    mods = build_modules(cfg, W)
    embed_w = W["llm"]["embed_tokens.weight"]
    lo = one_pass(mods, case, cfg, torch.bfloat16, embed_w, log)
    # isolated LLM floor: the SAME (bf16-valued) embeddings through the LLM in both precisions
    hi_iso = None
    hi = one_pass(mods, case, cfg, torch.float32, embed_w.float(), log)
    hi_iso = llm_isolated(mods, lo["embeds"], lo["pos"], embed_w.float(), torch.float32)
    fl = {}
    for k in ("vit_tokens", "vit_map", "image_tokens", "hfre_features", "region_tokens", "llm_hidden", "llm_final_last_row"):
        fl[k] = metrics(lo[k], hi[k])
    for i in range(4):
        fl[f"davit_stage{i}"] = metrics(lo[f"davit_stage{i}"], hi[f"davit_stage{i}"])
        fl[f"fpn_level{i}"] = metrics(lo[f"fpn_level{i}"], hi[f"fpn_level{i}"])
    fl["hfre_composed"] = fl["hfre_features"]
    fl["llm_hidden_composed"] = fl["llm_hidden"]
    fl["llm_hidden"] = metrics(lo["llm_hidden"], hi_iso[0])
    fl["llm_final_last_row_isolated"] = metrics(lo["llm_final_last_row"], hi_iso[1])
    fl["logits_composed"] = dict(max_abs=float((lo["logits"] - hi["logits"]).abs().max()), ref_std=float(hi["logits"].std()))
    fl["logits_isolated"] = dict(max_abs=float((lo["logits"] - hi_iso[2]).abs().max()), ref_std=float(hi_iso[2].std()))
    fl["_meta"] = dict(what="reference modules, bf16 execution vs fp32 execution, same weights and inputs, CPU", depth_vit=cfg.vit.depth,
                       layers_llm=cfg.llm.num_layers, torch=torch.__version__, boxes=100, image="640x480")
    name = "bf16_floor_small.json" if os.environ.get("FLOOR_SMALL") else "bf16_floor.json"
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), name), "w") as f:
        json.dump(fl, f, indent=1)
    for k, v in fl.items():
        print(k, v)


if __name__ == "__main__":
    main()
