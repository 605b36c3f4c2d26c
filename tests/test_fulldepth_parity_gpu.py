"""FULL-DEPTH parity on the metric's own configuration (BASELINE.json `metric`: 640x480 image x 100 proposals, Qwen2.5-VL-3B:
32-block ViT + DaViT-L + SimpleFPN + HFRE(100 boxes) + 36-layer LLM) — the HIP engine against the composed CPU oracle
(oracle/*.py, fp32 arithmetic on the same bf16-valued weights).  Replaces the reference call chain
`encode_images` -> `encode_regions` -> splice -> `Qwen2_5_VLModel.forward` (omchat_qwen2_5_vl.py:44-128,135-463,
modeling_qwen2_5_vl.py:1126-1242) end to end.

Two views per stage:
  * composed  — the oracle consumes ITS OWN upstream outputs: the error the engine accumulates over the whole path;
  * isolated  — the oracle stage consumes the ENGINE's (bf16) inputs: that stage's own error, no upstream drift.

Tolerances: SURVEY §7 asks for per-token cosine >= 0.9999 and max|d|/max|x| <= 2^-5 "after 32-36 layers" as a starting point
"to tighten after first measurements".  bf16 storage re-rounds every operator output; the reference's own bf16 execution
deviates from an fp32 evaluation of the same weights by the floors in tests/golden/bf16_floor.json, measured in the build
container by running the reference's own modules in bf16 and in fp32 on the CPU at full depth on this very configuration
(tests/golden/measure_bf16_floor.py; e.g. ViT map min cos 0.99967, LLM layer-36 hidden 0.99899 / rel 0.032, last-row logits
0.14).  Each assertion uses min(SURVEY bound, measured floor with a 1.5x margin on 1 - cos and on rel): the engine may be
as noisy as the reference's bf16 execution, not noisier.  First measurement on MI355X (profiles/r02_fulldepth_metrics_first.json):
the engine sits AT that floor at every stage (ViT map 0.99963, DaViT stage 3 0.99979, LLM hidden 0.99900, logits 0.13).
Every metric is also written to gpurun_out/fulldepth_metrics.json so the numbers behind the assertions are on record."""
import json
import os
import sys
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _floors():
    p = os.path.join(ROOT, "tests", "golden", "bf16_floor.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def metrics(got, ref):
    got, ref = got.float().cpu().reshape(-1, got.shape[-1]), ref.float().reshape(-1, ref.shape[-1])
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    cos = F.cosine_similarity(got, ref, dim=-1)
    return dict(min_cos=float(cos.min()), mean_cos=float(cos.mean()), rel=float((got - ref).abs().max() / ref.abs().max()),
                rms_rel=float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))


def mlp2(x, sd, prefix):
    h = F.gelu(F.linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))
    return F.linear(h, sd[prefix + "2.weight"], sd[prefix + "2.bias"])


@pytest.fixture(scope="module")
def run():
    """One engine pass + one composed oracle pass (about 10-20 s of host time on 32 threads), shared by the assertions."""
    import bench
    from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, vit_oracle as VO
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda", 0)
    case = bench.build_workload(dev, n_boxes=100, seed=77)
    pipe = bench.Pipeline(case, dev, inflight=1)
    eng, cfg = pipe.eng, pipe.cfg
    assert cfg.vit.depth == 32 and cfg.llm.num_layers == 36 and case["boxes"].shape[0] == 100
    gh, gw = case["grid"]
    H, W = case["img_hw"]
    d = case["dev"]
    # ---- engine: the product path, plus its intermediates via the sub-modules (same kernels, eager) ----
    out = eng.prefill(case["ids"], d["pix"], (gh, gw), d["aux"], d["boxes"], use_graph=False)
    out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
    e_tokens, e_feats = eng.vit.forward(d["pix"], gh, gw, capture="last")
    e_aux, e_aux_sizes = eng.davit.forward(d["aux"])
    e_fpn, e_fpn_sizes = eng.fpn.forward(e_feats[-1], gh, gw)
    coll = []
    eng.llm.prefill(out["embeds"], out["position_ids"], out["rope_delta"], collect=coll)
    e_hidden = coll[-1].clone()
    torch.cuda.synchronize()

    def nchw(t, hw):
        return t.view(1, hw[0], hw[1], t.shape[1]).permute(0, 3, 1, 2)

    sw, sh = gw * 14 / W, gh * 14 / H
    boxes = case["boxes"]
    vtb = boxes * torch.tensor([sw, sh, sw, sh])
    eng.hfre.simple_fpn = lambda x: [nchw(t, s) for t, s in zip(e_fpn, e_fpn_sizes)]
    e_feat = eng.hfre([nchw(t, s) for t, s in zip(e_aux, e_aux_sizes)], [d["boxes"]], nchw(e_feats[-1], (gh, gw)), None,
                      vt_scale=(sw, sh))[0].clone()
    torch.cuda.synchronize()
    # ---- oracle, composed ----
    t0 = time.perf_counter()
    sd = {k: {n: t.float().cpu() for n, t in v.items()} for k, v in pipe.weights.items()}
    o_tokens, o_maps = VO.vit_forward(sd["vit"], case["pix"].float(), gh, gw, depth=32, n_heads=16, fullatt=(7, 15, 23, 31))
    o_img = mlp2(o_tokens, sd["proj"], "mm_projector.")
    o_fpn = FO.fpn_forward(sd["fpn"], o_maps[-1].bfloat16().float().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
    o_aux, o_aux_sizes = DO.davit_forward(sd["davit"], case["aux"].float().unsqueeze(0))
    o_aux_nchw = [m.bfloat16().reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(o_aux, o_aux_sizes)]
    o_feat = HO.hfre_oracle(o_aux_nchw, boxes, [m.bfloat16() for m in o_fpn], vtb, region_dim=5888, grid_hw=(gh, gw),
                            vt_strides=[3.5, 7, 14, 28])[0]
    o_reg = mlp2(o_feat.bfloat16().float(), sd["proj"], "mm_projector_aux.")
    o_emb, nb, na = LO.splice(torch.tensor(case["ids"]), sd["llm"]["embed_tokens.weight"], o_img, o_reg)
    o_pos, o_delta = LO.rope_index(nb, (gh // 2, gw // 2), na)
    kw = dict(n_layers=36, n_heads=16, n_kv=2, head_dim=128, eps=1e-6, theta=1e6, sections=(16, 24, 24))
    o_final, o_hs = LO.llm_forward(sd["llm"], o_emb, o_pos, return_all=True, **kw)
    o_logits = o_final[-1:] @ sd["llm"]["embed_tokens.weight"].t()
    t_oracle = time.perf_counter() - t0
    # ---- oracle, isolated: each stage on the engine's own inputs ----
    i_fpn = FO.fpn_forward(sd["fpn"], e_feats[-1].float().cpu().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
    i_feat = HO.hfre_oracle([nchw(t, s).cpu() for t, s in zip(e_aux, e_aux_sizes)], boxes, [nchw(t, s).cpu() for t, s in zip(e_fpn, e_fpn_sizes)],
                            vtb, region_dim=5888, grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
    i_reg = mlp2(e_feat.bfloat16().float().cpu(), sd["proj"], "mm_projector_aux.")
    i_final, i_hs = LO.llm_forward(sd["llm"], out["embeds"].float().cpu(), out["position_ids"], return_all=True, **kw)
    i_logits = i_final[-1:] @ sd["llm"]["embed_tokens.weight"].t()

    def tm(m):   # oracle NCHW fp32 map -> token-major
        return m[0].permute(1, 2, 0).reshape(-1, m.shape[1])

    M = {}
    M["vit_image_tokens(32 blocks+merger)"] = metrics(e_tokens, o_tokens)
    M["vit_last_fullatt_map(block 31)"] = metrics(e_feats[-1], o_maps[-1])
    M["image_tokens(mm_projector)"] = metrics(out["image_tokens"], o_img)
    for i in range(4):
        M[f"davit_stage{i}"] = metrics(e_aux[i], o_aux[i])
        M[f"fpn_level{i}_composed"] = metrics(e_fpn[i], tm(o_fpn[i]))
        M[f"fpn_level{i}_isolated"] = metrics(e_fpn[i], tm(i_fpn[i]))
    M["hfre_features_composed"] = metrics(e_feat, o_feat)
    M["hfre_features_isolated"] = metrics(e_feat, i_feat)
    M["hfre_features_isolated"]["max_abs"] = float((e_feat.float().cpu() - i_feat).abs().max())
    M["region_tokens_composed"] = metrics(out["region_tokens"], o_reg)
    M["region_tokens_isolated"] = metrics(out["region_tokens"], i_reg)
    M["llm_hidden_layer36_composed"] = metrics(e_hidden, o_hs[-1])
    M["llm_hidden_layer36_isolated"] = metrics(e_hidden, i_hs[-1])
    M["llm_last_row_final_norm_composed"] = metrics(out["last_hidden"], o_final[-1:])
    M["llm_last_row_final_norm_isolated"] = metrics(out["last_hidden"], i_final[-1:])
    lg = out["logits"].float().cpu()
    for name, ref in (("composed", o_logits), ("isolated", i_logits)):
        top2 = ref[0].topk(2).values
        M[f"logits_{name}"] = dict(max_abs=float((lg - ref).abs().max()), ref_std=float(ref.std()), margin=float(top2[0] - top2[1]),
                                   argmax_equal=bool(int(out["next_token"].item()) == int(ref.argmax())))
    M["_meta"] = dict(oracle_seconds=round(t_oracle, 2), threads=torch.get_num_threads(), L=int(out["embeds"].shape[0]),
                      position_ids_equal=bool(torch.equal(o_pos, out["position_ids"])), rope_delta_equal=bool(o_delta == out["rope_delta"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fulldepth_metrics.json"), "w") as f:
        json.dump(M, f, indent=1)
    return M


def bound(stage, survey_cos=0.9999, survey_rel=2 ** -5):
    """(min cosine, max rel) for a stage: the SURVEY §7 bound, relaxed to the measured reference-bf16 floor (x1.5 on the
    deviation) where the reference's own bf16 execution is already worse than the SURVEY starting point."""
    fl = _floors().get(stage)
    if not fl:
        return survey_cos, survey_rel
    return min(survey_cos, 1.0 - 1.5 * (1.0 - fl["min_cos"])), max(survey_rel, 1.5 * fl["rel"])


def check(M, key, stage, **kw):
    cmin, rmax = bound(stage, **kw)
    m = M[key]
    assert m["min_cos"] >= cmin and m["rel"] <= rmax, f"{key}: min cos {m['min_cos']:.6f} (need {cmin:.6f}), rel {m['rel']:.4g} (need {rmax:.4g})"


def test_index_bookkeeping_exact(run):
    assert run["_meta"]["position_ids_equal"] and run["_meta"]["rope_delta_equal"] and run["_meta"]["L"] == 651


def test_vit_32_blocks(run):
    check(run, "vit_last_fullatt_map(block 31)", "vit_map")
    check(run, "vit_image_tokens(32 blocks+merger)", "vit_tokens")
    check(run, "image_tokens(mm_projector)", "image_tokens")


def test_davit_large(run):
    for i in range(4):
        check(run, f"davit_stage{i}", f"davit_stage{i}")


def test_simple_fpn(run):
    for i in range(4):
        check(run, f"fpn_level{i}_isolated", "fpn", survey_cos=0.9998)
        check(run, f"fpn_level{i}_composed", f"fpn_level{i}")


def test_hfre_100_boxes(run):
    # the north-star kernel: fp32 out on identical bf16 inputs -> the HFRE tolerance of tests/test_hfre_gpu.py
    m = run["hfre_features_isolated"]
    assert m["min_cos"] >= 0.999999 and m["rel"] <= 1e-4, f"hfre isolated: {m}"
    check(run, "hfre_features_composed", "hfre_composed")


def test_region_tokens(run):
    # north_star: "region-token tensors match the reference within a stated bf16 tolerance"
    m = run["region_tokens_isolated"]
    assert m["min_cos"] >= 0.9999 and m["rel"] <= 2 ** -6, f"region tokens (connector alone): {m}"
    check(run, "region_tokens_composed", "region_tokens")


def test_llm_36_layers(run):
    check(run, "llm_hidden_layer36_isolated", "llm_hidden")
    check(run, "llm_last_row_final_norm_isolated", "llm_hidden")
    check(run, "llm_hidden_layer36_composed", "llm_hidden_composed")


def test_logits_and_first_token(run):
    for name in ("isolated", "composed"):
        m = run[f"logits_{name}"]
        fl = _floors().get(f"logits_{name}", {}).get("max_abs", 0.0)
        tol = max(0.05, 1.5 * fl)
        assert m["max_abs"] <= tol, f"logits {name}: max|d| {m['max_abs']:.4g} > {tol:.4g}"
        if m["margin"] > 2 * tol:
            assert m["argmax_equal"], f"first greedy token differs from the oracle's ({name}) although its margin {m['margin']:.3g} > {2 * tol:.3g}"
