"""FULL-DEPTH parity (32-block ViT + DaViT-L + SimpleFPN + HFRE + 36-layer LLM + greedy decode) of the HIP engine on three BASELINE
configurations (tests/fulldepth_case.py): `metric` (640x480 x 100 proposals — the configuration the metric is quoted on), `demo`
(configs[0]: 500x399 x the 7 boxes of inference.py:16) and `hires` (configs[4]: 1344x1344 x 300 proposals as 3 prompts of 100 — the
reference caps features at 100 per prompt, mm_utils.py:600).  Replaces the reference call chain `encode_images` -> `encode_regions` ->
splice -> `Qwen2_5_VLModel.forward` -> greedy decode (omchat_qwen2_5_vl.py:44-128,135-463,143-155; modeling_qwen2_5_vl.py:1126-1242).

Two checkers:
  * the composed CPU oracle (oracle/*.py, fp32 on the same bf16-valued weights), evaluated on this box — all three cases, two views per
    stage: composed (the oracle consumes ITS OWN upstream outputs: accumulated error) and isolated (the oracle stage consumes the
    ENGINE's inputs: that stage's own error);
  * outputs of the REFERENCE's own modules at full depth, generated in the build container (tests/golden/make_fulldepth_ref.py ->
    fulldepth_ref_{metric,demo}.npz): fp32 execution and, for `metric`, the reference's bf16 execution (`model.to(bfloat16)`) —
    region tokens, last hidden state, K = 16 greedy ids with their top-8 logits.  Weights are seeded on the CPU on both machines and
    checksummed.

Tolerances.  bf16 storage re-rounds every operator output; the reference's own bf16 execution deviates from an fp32 evaluation of the
same weights by the floors in tests/golden/bf16_floor.json (reference modules run in bf16 and fp32 at full depth on the metric
configuration, tests/golden/measure_bf16_floor.py).  Each stage assertion uses min(SURVEY §7 bound, floor x 1.5): the engine may be as
noisy as the reference's bf16 execution, not noisier.  Against the reference-bf16 golden the bound is the triangle one: two bf16
executions may each sit at the floor, so their mutual deviation is bounded by floor(engine) + floor(reference) <= 2.5 x floor.
DECODED IDS (north_star: "decoded text/box-index outputs are bit-identical"): K = 16 teacher-forced greedy steps at 36 layers on a
peaked test head (tests/fulldepth_case.py).  What decides an argmax is the error of logit DIFFERENCES, and it is measured, not guessed:
sigma = the rms of |(top1 - top_j)_bf16 - (top1 - top_j)_fp32| over the reference's own two executions (golden, K steps x 7 pairs).
At every step whose reference margin exceeds 3 sigma the engine's id must EQUAL the reference's; the engine's own difference error
must stay <= 1.5 sigma (rms) and <= 3 sigma on every top-1 margin; the engine must agree with the fp32 reference on at least as many
steps as the reference's own bf16 execution does, minus one; and the free-running device-loop ids must be the reference's ids up to
the first unqualified step.  With seeded random weights the last hidden state carries ~4-6 % of bf16 noise after 36 layers, the same
order as the gap between random competitors, so a share of the steps is a coin flip for ANY bf16 execution — the reference's own
bf16 run disagrees with its fp32 run on 3 of the 16 metric steps.  The golden has 7 / 16 (metric) and 16 / 16 (demo) qualified steps;
the test requires >= K/3.  (Round 2 compared against the vocabulary-wide maximum logit error — an extreme-value statistic over 152k
entries, ~4x the per-entry noise — with an iid head: no step ever qualified, VERDICT r2 weak #1.)
Every metric is written to gpurun_out/fulldepth_metrics_<case>.json (copied to profiles/)."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _floors():
    p = os.path.join(GOLD, "bf16_floor.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def metrics(got, ref):
    got, ref = got.float().cpu().reshape(-1, got.shape[-1]), ref.float().reshape(-1, ref.shape[-1])
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    cos = F.cosine_similarity(got, ref, dim=-1)
    return dict(min_cos=float(cos.min()), mean_cos=float(cos.mean()), rel=float((got - ref).abs().max() / ref.abs().max()),
                rms_rel=float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))


@pytest.fixture(scope="module")
def world():
    """CPU-seeded weights at the true shapes (~40 s), one engine, the CPU fp32 state for the oracle."""
    import composed_oracle as CO
    import fulldepth_case as FC
    from vlm_fo1_amd.model import FO1Engine
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    cfg = FC.full_config()
    W, cks = FC.cpu_weights(cfg)
    dev = torch.device("cuda", 0)
    eng = FO1Engine(cfg, {k: {n: t.to(dev) for n, t in v.items()} for k, v in W.items()}, dev)
    assert eng.llm.lm_head.data_ptr() != eng.llm.embed.data_ptr(), "the untied test head must be in use"
    sd = CO.cpu_state(W)
    del W
    return dict(cfg=cfg, eng=eng, sd=sd, cks=cks, dev=dev, cache={})


def golden(name, cks):
    p = os.path.join(GOLD, f"fulldepth_ref_{name}.npz")
    if not os.path.exists(p):
        return None
    g = np.load(p)
    same = all(int(g[f"cks_{k}"]) == v for k, v in cks.items())
    assert same, ("the CPU random stream on this box does not reproduce the weights the reference golden was made with "
                  f"(checksums {cks} vs {[int(g['cks_' + k]) for k in cks]}): regenerate tests/golden/fulldepth_ref_*.npz")
    return g


def pair_err(got8, ref8):
    """|(x0 - xj)_got - (x0 - xj)_ref| for j = 1..7 over the reference's top-8 entries: the error of the differences an argmax compares."""
    got8, ref8 = got8.double(), ref8.double()
    return ((got8[:1] - got8[1:]) - (ref8[:1] - ref8[1:])).abs().tolist()


def engine_decode_forced(eng, req, forced, K):
    """Prefill + K - 1 teacher-forced decode steps (eager launches, the same device code the graph replays).  -> (ids the engine
    would have picked [K], logits [K, V] fp32 on the host)."""
    out = eng.prefill(req["ids"], req["pix"], req["grid"], req["aux"], req["boxes"], use_graph=False)
    eng.llm.reserve(eng.llm.kv_len + K + 1)
    ids, logits = [int(out["next_token"].item())], [out["logits"][0].float().cpu()]
    for i in range(K - 1):
        tok = torch.tensor([int(forced[i])], dtype=torch.int32, device=eng.dev)
        _, lg, nxt = eng.llm.decode_step(tok)
        ids.append(int(nxt.item()))
        logits.append(lg[0].float().cpu())
    return ids, torch.stack(logits)


def run_case(world, name):
    if name in world["cache"]:
        return world["cache"][name]
    import composed_oracle as CO
    import fulldepth_case as FC
    from oracle import llm_oracle as LO
    cfg, eng, sd, dev = world["cfg"], world["eng"], world["sd"], world["dev"]
    case = FC.build_case(name)
    gh, gw = case["grid"]
    H, W = case["img_hw"]
    K = FC.K_DECODE if name != "hires" else 8
    pix, aux = case["pix"].to(dev), case["aux"].to(dev)
    reqs = [dict(ids=ids, pix=pix, grid=(gh, gw), aux=aux, boxes=b.to(dev)) for ids, b in case["groups"]]
    M = {}
    # ---- engine: the product path (one packed pass over the prompts of the case), then its intermediates via the sub-modules ----
    t0 = time.perf_counter()
    outs = eng.prefill_batch(reqs, use_graph=False)
    outs = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in outs]
    out = outs[0]
    e_tokens, e_feats = eng.vit.forward(pix, gh, gw, capture="last")
    e_aux, e_aux_sizes = eng.davit.forward(aux)
    e_fpn, e_fpn_sizes = eng.fpn.forward(e_feats[-1], gh, gw)
    e_tokens, e_feats = e_tokens.clone(), [t.clone() for t in e_feats]
    e_aux, e_fpn = [t.clone() for t in e_aux], [t.clone() for t in e_fpn]

    def nchw(t, hw):
        return t.view(1, hw[0], hw[1], t.shape[1]).permute(0, 3, 1, 2)

    sw, sh = gw * 14 / W, gh * 14 / H
    boxes_all = case["boxes"]
    eng.hfre.simple_fpn = lambda x: [nchw(t, s) for t, s in zip(e_fpn, e_fpn_sizes)]
    e_feat = eng.hfre([nchw(t, s) for t, s in zip(e_aux, e_aux_sizes)], [boxes_all.to(dev)], nchw(e_feats[-1], (gh, gw)), None,
                      vt_scale=(sw, sh))[0].clone()
    coll = []
    eng.llm.prefill(out["embeds"], out["position_ids"], out["rope_delta"], collect=coll)
    e_hidden = coll[-1].clone()
    del coll
    torch.cuda.synchronize()
    # free-running greedy ids: device loop (BatchDecoder, graph) for every prompt of the case, and the host loop for prompt 0
    free_batch = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)
    free_host = eng.generate(reqs[0]["ids"], pix, (gh, gw), aux, reqs[0]["boxes"], max_new_tokens=K, use_graph=True)
    t_engine = time.perf_counter() - t0
    # ---- oracle, composed ----
    t0 = time.perf_counter()
    o_tokens, o_maps = CO.vit(sd, case["pix"], gh, gw, cfg.vit)
    o_img = CO.projector(o_tokens, sd["proj"], "mm_projector.", cfg.mm_projector_type)
    o_aux = CO.davit_maps(sd, case["aux"])
    from oracle import fpn_oracle as FO, hfre_oracle as HO
    o_fpn = FO.fpn_forward(sd["fpn"], CO.nchw(o_maps[-1].bfloat16().float(), gh, gw))
    vtb = boxes_all * torch.tensor([sw, sh, sw, sh])
    o_feat = HO.hfre_oracle(o_aux, boxes_all, [m.bfloat16() for m in o_fpn], vtb, region_dim=5888, grid_hw=(gh, gw), vt_strides=CO.FPN_STRIDES)[0]
    o_reg = CO.region_tokens(sd, cfg, o_feat)
    n0 = case["groups"][0][1].shape[0]
    o_llm = CO.llm_prefill(sd, cfg, case["groups"][0][0], o_img, o_reg[:n0], gh, gw, return_all=True)
    # ---- oracle, isolated: each stage on the engine's own inputs ----
    i_fpn = FO.fpn_forward(sd["fpn"], CO.nchw(e_feats[-1].float().cpu(), gh, gw))
    i_feat = HO.hfre_oracle([nchw(t, s).cpu() for t, s in zip(e_aux, e_aux_sizes)], boxes_all, [nchw(t, s).cpu() for t, s in zip(e_fpn, e_fpn_sizes)],
                            vtb, region_dim=5888, grid_hw=(gh, gw), vt_strides=CO.FPN_STRIDES)[0]
    i_reg = CO.region_tokens(sd, cfg, e_feat.cpu())
    i_final, i_hs = LO.llm_forward(sd["llm"], out["embeds"].float().cpu(), out["position_ids"], return_all=True, **o_llm["kw"])
    head = sd["llm"]["lm_head.weight"]
    i_logits = i_final[-1:] @ head.t()
    # ---- decoded ids: the oracle teacher-forced on the ENGINE's free-running ids (every step comparable) ----
    forced = free_batch[0]
    e_ids, e_logits = engine_decode_forced(eng, reqs[0], forced, K)
    ref_ids, ref_logits = LO.greedy_decode(sd["llm"], out["embeds"].float().cpu(), out["position_ids"], out["rope_delta"], K,
                                           lm_head=head, forced=forced, **o_llm["kw"])
    t_oracle = time.perf_counter() - t0

    def tm(m):   # oracle NCHW fp32 map -> token-major
        return m[0].permute(1, 2, 0).reshape(-1, m.shape[1])

    M["vit_image_tokens(32 blocks+merger)"] = metrics(e_tokens, o_tokens)
    M["vit_last_fullatt_map(block 31)"] = metrics(e_feats[-1], o_maps[-1])
    M["image_tokens(mm_projector)"] = metrics(out["image_tokens"], o_img)
    for i in range(4):
        M[f"davit_stage{i}"] = metrics(e_aux[i], tm(o_aux[i].float()))
        M[f"fpn_level{i}_composed"] = metrics(e_fpn[i], tm(o_fpn[i]))
        M[f"fpn_level{i}_isolated"] = metrics(e_fpn[i], tm(i_fpn[i]))
    M["hfre_features_composed"] = metrics(e_feat, o_feat)
    M["hfre_features_isolated"] = metrics(e_feat, i_feat)
    M["hfre_features_isolated"]["max_abs"] = float((e_feat.float().cpu() - i_feat).abs().max())
    e_reg = torch.cat([o["region_tokens"] for o in outs])            # the packed pass's region tokens, prompt by prompt
    M["region_tokens_composed"] = metrics(e_reg, o_reg)
    M["region_tokens_isolated"] = metrics(e_reg, i_reg)
    M["llm_hidden_layer36_composed"] = metrics(e_hidden, o_llm["hidden"][-1])
    M["llm_hidden_layer36_isolated"] = metrics(e_hidden, i_hs[-1])
    M["llm_last_row_final_norm_composed"] = metrics(out["last_hidden"], o_llm["final"][-1:])
    M["llm_last_row_final_norm_isolated"] = metrics(out["last_hidden"], i_final[-1:])
    lg = out["logits"].float().cpu()
    for nm, ref in (("composed", o_llm["logits"]), ("isolated", i_logits)):
        top = ref[0].topk(8)
        M[f"logits_{nm}"] = dict(max_abs_whole_vocab=float((lg - ref).abs().max()), max_abs_top8=float((lg[0][top.indices] - top.values).abs().max()),
                                 ref_std=float(ref.std()), margin=float(top.values[0] - top.values[1]),
                                 argmax_equal=bool(int(out["next_token"].item()) == int(ref.argmax())))
    steps = []
    for i in range(K):
        top = ref_logits[i].topk(8)
        steps.append(dict(engine_id=e_ids[i], oracle_id=ref_ids[i], margin=float(top.values[0] - top.values[1]),
                          err_top8=float((e_logits[i][top.indices] - top.values).abs().max()), pair_err=pair_err(e_logits[i][top.indices], top.values),
                          engine_token_deficit=float(ref_logits[i].max() - ref_logits[i][e_ids[i]])))
    M["decode_vs_oracle"] = dict(K=K, forced_on="engine free-running ids", steps=steps, free_batch=free_batch, free_host=free_host,
                                 engine_forced_ids=e_ids)
    M["_meta"] = dict(case=name, image=f"{W}x{H}", grid=[gh, gw], prompts=len(reqs), boxes=int(boxes_all.shape[0]), L=int(out["embeds"].shape[0]),
                      oracle_seconds=round(t_oracle, 1), engine_seconds=round(t_engine, 1), threads=torch.get_num_threads(),
                      position_ids_equal=bool(torch.equal(o_llm["pos"], out["position_ids"])), rope_delta_equal=bool(o_llm["delta"] == out["rope_delta"]))
    # ---- the reference's own outputs (build-container golden) ----
    g = golden(name, world["cks"])
    if g is not None:
        ref_fp32_ids = g["fp32_ids"].tolist()
        r_ids, r_logits = engine_decode_forced(eng, reqs[0], ref_fp32_ids, K)
        R = dict(fp32_ids=ref_fp32_ids, engine_forced_on_fp32_ids=r_ids)
        R["region_tokens_vs_fp32"] = metrics(outs[0]["region_tokens"], torch.from_numpy(g["fp32_region_tokens"].astype(np.float32)))
        R["last_hidden_vs_fp32"] = metrics(out["last_hidden"], torch.from_numpy(g["fp32_last_hidden"]))
        R["image_tokens_rows_vs_fp32"] = metrics(out["image_tokens"][::8], torch.from_numpy(g["fp32_image_tokens_rows"].astype(np.float32)))
        R["hfre_rows_vs_fp32"] = metrics(e_feat[:8], torch.from_numpy(g["fp32_hfre_rows"]))
        top_i, top_v = torch.from_numpy(g["fp32_top_ids"]), torch.from_numpy(g["fp32_top_vals"])
        R["steps"] = [dict(margin=float(top_v[i, 0] - top_v[i, 1]), err_top8=float((r_logits[i][top_i[i]] - top_v[i]).abs().max()),
                           pair_err=pair_err(r_logits[i][top_i[i]], top_v[i])) for i in range(K)]
        if "bf16_ids" in g.files:
            R["bf16_ids"] = g["bf16_ids"].tolist()
            R["floor_top8"] = float(np.abs(g["bf16_top_vals"] - g["fp32_top_vals"]).max())     # reference bf16 vs fp32 at the fp32 top-8, all K steps
            R["region_tokens_vs_bf16"] = metrics(outs[0]["region_tokens"], torch.from_numpy(g["bf16_region_tokens"].astype(np.float32)))
            R["last_hidden_vs_bf16"] = metrics(out["last_hidden"], torch.from_numpy(g["bf16_last_hidden"]))
            R["ref_bf16_vs_fp32"] = dict(region_tokens=metrics(torch.from_numpy(g["bf16_region_tokens"].astype(np.float32)),
                                                               torch.from_numpy(g["fp32_region_tokens"].astype(np.float32))),
                                         last_hidden=metrics(torch.from_numpy(g["bf16_last_hidden"]), torch.from_numpy(g["fp32_last_hidden"])))
        M["vs_reference_golden"] = R
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fulldepth_metrics_{name}.json"), "w") as f:
        json.dump(M, f, indent=1)
    world["cache"][name] = M
    return M


def bound(stage, survey_cos=0.9999, survey_rel=2 ** -5, slack=1.5):
    """(min cosine, max rel) for a stage: the SURVEY §7 bound, relaxed to the measured reference-bf16 floor (x slack on the
    deviation) where the reference's own bf16 execution is already worse than the SURVEY starting point."""
    fl = _floors().get(stage)
    if not fl:
        return survey_cos, survey_rel
    return min(survey_cos, 1.0 - slack * (1.0 - fl["min_cos"])), max(survey_rel, slack * fl["rel"])


def check(M, key, stage, **kw):
    cmin, rmax = bound(stage, **kw)
    m = M[key]
    assert m["min_cos"] >= cmin and m["rel"] <= rmax, f"{key}: min cos {m['min_cos']:.6f} (need {cmin:.6f}), rel {m['rel']:.4g} (need {rmax:.4g})"


def noise():
    """sigma of the logit-difference error of the reference's OWN bf16 execution against its fp32 execution (metric golden): rms over
    K steps x 7 pairs of |(top1 - top_j)_bf16 - (top1 - top_j)_fp32|."""
    p = os.path.join(GOLD, "fulldepth_ref_metric.npz")
    assert os.path.exists(p), "tests/golden/fulldepth_ref_metric.npz is missing (tests/golden/make_fulldepth_ref.py)"
    g = np.load(p)
    a, b = g["bf16_top_vals"].astype(np.float64), g["fp32_top_vals"].astype(np.float64)
    d = (a[:, :1] - a[:, 1:]) - (b[:, :1] - b[:, 1:])
    return float(np.sqrt((d ** 2).mean()))


def check_decode(name, steps, ids_engine, ids_ref, K):
    sig = noise()
    qualified = agree = 0
    allp = []
    for i, s in enumerate(steps):
        allp += s["pair_err"]
        same = ids_engine[i] == ids_ref[i]
        agree += int(same)
        assert s["pair_err"][0] <= 3.0 * sig, f"{name} step {i}: the engine's top-1 margin is off by {s['pair_err'][0]:.3g} (> 3 sigma = {3.0 * sig:.3g})"
        if s["margin"] > 3.0 * sig:
            qualified += 1
            assert same, f"{name} step {i}: engine id {ids_engine[i]} != reference id {ids_ref[i]} at margin {s['margin']:.3g} > 3 sigma = {3.0 * sig:.3g}"
    rms = float(np.sqrt(np.mean(np.square(allp))))
    assert rms <= 1.5 * sig, f"{name}: logit-difference error rms {rms:.3g} > 1.5 x the reference's own bf16 noise {sig:.3g}"
    assert qualified >= K // 3, f"{name}: only {qualified} of {K} steps have a margin > 3 sigma = {3.0 * sig:.3g}: the test is not discriminating"
    return agree


CASES = ["metric", "demo", "hires"]


@pytest.mark.parametrize("name", CASES)
def test_index_bookkeeping_exact(world, name, product_library):
    M = run_case(world, name)
    assert M["_meta"]["position_ids_equal"] and M["_meta"]["rope_delta_equal"]
    if name == "metric":
        assert M["_meta"]["L"] == 651


@pytest.mark.parametrize("name", CASES)
def test_towers_full_depth(world, name, product_library):
    M = run_case(world, name)
    check(M, "vit_last_fullatt_map(block 31)", "vit_map")
    check(M, "vit_image_tokens(32 blocks+merger)", "vit_tokens")
    check(M, "image_tokens(mm_projector)", "image_tokens")
    for i in range(4):
        check(M, f"davit_stage{i}", f"davit_stage{i}")
        check(M, f"fpn_level{i}_isolated", "fpn", survey_cos=0.9998)
        check(M, f"fpn_level{i}_composed", f"fpn_level{i}")


@pytest.mark.parametrize("name", CASES)
def test_hfre_and_region_tokens(world, name, product_library):
    M = run_case(world, name)
    # the north-star kernel: fp32 out on identical bf16 inputs -> the HFRE tolerance of tests/test_hfre_gpu.py
    m = M["hfre_features_isolated"]
    assert m["min_cos"] >= 0.999999 and m["rel"] <= 1e-4, f"hfre isolated: {m}"
    check(M, "hfre_features_composed", "hfre_composed")
    # north_star: "region-token tensors match the reference within a stated bf16 tolerance"
    m = M["region_tokens_isolated"]
    assert m["min_cos"] >= 0.9999 and m["rel"] <= 2 ** -6, f"region tokens (connector alone): {m}"
    check(M, "region_tokens_composed", "region_tokens")


@pytest.mark.parametrize("name", CASES)
def test_llm_36_layers(world, name, product_library):
    M = run_case(world, name)
    check(M, "llm_hidden_layer36_isolated", "llm_hidden")
    check(M, "llm_last_row_final_norm_isolated", "llm_hidden")
    check(M, "llm_hidden_layer36_composed", "llm_hidden_composed")


@pytest.mark.parametrize("name", CASES)
def test_decoded_ids_vs_oracle(world, name, product_library):
    """K teacher-forced greedy steps at 36 layers against the oracle's KV-cache decode (pinned to the reference's vendored model)."""
    M = run_case(world, name)
    D = M["decode_vs_oracle"]
    check_decode(name, D["steps"], [s["engine_id"] for s in D["steps"]], [s["oracle_id"] for s in D["steps"]], D["K"])
    # Three engine decode paths — device loop (BatchDecoder: MFMA skinny GEMM), host loop (one-sequence GEMV graph) and the eager
    # teacher-forced steps — use kernels with different fp32 summation orders: they must agree up to the first step whose oracle margin
    # is a near-tie (< 3 sigma), where a different rounding may legitimately pick the other candidate (first run: metric step 11,
    # margin 0.13 against sigma 0.59 — the device loop took one candidate, the host loop the other, the fp32 reference the first).
    sig = noise()
    n_ok = next((i for i, s in enumerate(D["steps"]) if s["margin"] <= 3.0 * sig), D["K"])
    assert D["free_batch"][0][:n_ok + 1][:n_ok] == D["free_host"][:n_ok], "device-loop and host-loop greedy decodes differ before any near-tie"
    assert D["engine_forced_ids"][:n_ok] == D["free_batch"][0][:n_ok], "teacher-forcing the engine on its own ids must reproduce them"


@pytest.mark.parametrize("name", ["metric", "demo", "hires"])
def test_vs_reference_modules_at_full_depth(world, name, product_library):
    """The REFERENCE's own modules (build-container golden): ids, region tokens, last hidden state; and, for `metric` and `demo`, the
    engine against the reference's bf16 execution.  `hires` (configs[4]'s geometry): prompt 0 of the three prompts over the one image,
    fp32 reference (its towers run once for all three in the engine, once per prompt in the reference: same numbers)."""
    M = run_case(world, name)
    R = M.get("vs_reference_golden")
    assert R is not None, f"tests/golden/fulldepth_ref_{name}.npz is missing"
    K = len(R["fp32_ids"])
    check(dict(x=R["region_tokens_vs_fp32"]), "x", "region_tokens")
    check(dict(x=R["last_hidden_vs_fp32"]), "x", "llm_final_last_row", slack=2.0)       # one row: a single-sample statistic
    check(dict(x=R["image_tokens_rows_vs_fp32"]), "x", "image_tokens")
    check(dict(x=R["hfre_rows_vs_fp32"]), "x", "hfre_composed")
    agree = check_decode(name, R["steps"], R["engine_forced_on_fp32_ids"], R["fp32_ids"], K)
    if "bf16_ids" in R:
        ref_agree = sum(int(a == b) for a, b in zip(R["bf16_ids"], R["fp32_ids"]))
        assert agree >= ref_agree - 1, f"engine agrees with the fp32 reference on {agree}/{K} ids, the reference's own bf16 execution on {ref_agree}/{K}"
        # engine-bf16 vs reference-bf16: both may sit at the floor, so the mutual bound is the triangle one
        check(dict(x=R["region_tokens_vs_bf16"]), "x", "region_tokens", slack=2.5)
        check(dict(x=R["last_hidden_vs_bf16"]), "x", "llm_final_last_row", slack=3.0)
    # free-running: as long as the reference's own margins qualify, the engine's device-loop ids ARE the reference's ids
    sig = noise()
    free = M["decode_vs_oracle"]["free_batch"][0]
    for i, s in enumerate(R["steps"]):
        if s["margin"] <= 3.0 * sig:
            break
        assert free[i] == R["fp32_ids"][i], f"{name}: free-running id {free[i]} != the reference's {R['fp32_ids'][i]} at step {i}"


def test_fp8_presets_at_hires_against_the_reference_golden(world, product_library):
    """BASELINE configs[4] names fp8 MFMA; the reference has no fp8 path, so the fp8 engine's parity is UNPINNED by construction — this test
    BOUNDS it: the engine at configs[4]'s geometry (1344 x 1344, 300 proposals as 3 prompts of 100 over one image, shared
    prefix) with the e4m3 linears of every preset against the REFERENCE MODULES' fp32 golden (tests/golden/fulldepth_ref_hires.npz): region
    tokens, image tokens, last hidden state, and the decoded ids teacher-forced on the reference's ids with the margin criterion of the bf16
    test (3 sigma of the reference's own bf16-vs-fp32 logit-difference error).  The table goes to gpurun_out/r05_fp8_hires_metrics.json
    (committed as profiles/r05_fp8_hires_metrics.json); bench.py's hires.fp8 reports the preset named there."""
    import fulldepth_case as FC
    eng, dev = world["eng"], world["dev"]
    g = golden("hires", world["cks"])
    assert g is not None
    case = FC.build_case("hires")
    gh, gw = case["grid"]
    pix, aux = case["pix"].to(dev), case["aux"].to(dev)
    reqs = [dict(ids=ids, pix=pix, grid=(gh, gw), aux=aux, boxes=b.to(dev), image_id=0) for ids, b in case["groups"]]
    ref_ids = g["fp32_ids"].tolist()
    K = len(ref_ids)
    top_i, top_v = torch.from_numpy(g["fp32_top_ids"]), torch.from_numpy(g["fp32_top_vals"])
    sig = noise()
    table = {}
    for preset in (None, "llm-mlp", "mlp", "all"):
        n = eng.enable_fp8(preset) if preset else 0
        try:
            outs = eng.prefill_batch(reqs, use_graph=False)
            out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in outs[0].items()}
            r_ids, r_logits = engine_decode_forced(eng, reqs[0], ref_ids, K)
        finally:
            if preset:
                eng.disable_fp8()
        steps = [dict(margin=float(top_v[i, 0] - top_v[i, 1]), pair_err=pair_err(r_logits[i][top_i[i]], top_v[i])) for i in range(K)]
        qual = [i for i in range(K) if steps[i]["margin"] > 3.0 * sig]
        row = dict(fp8_weights=n,
                   region_tokens_vs_fp32=metrics(out["region_tokens"], torch.from_numpy(g["fp32_region_tokens"].astype(np.float32))),
                   image_tokens_rows_vs_fp32=metrics(out["image_tokens"][::8], torch.from_numpy(g["fp32_image_tokens_rows"].astype(np.float32))),
                   last_hidden_vs_fp32=metrics(out["last_hidden"], torch.from_numpy(g["fp32_last_hidden"])),
                   ids_equal_to_reference=sum(int(a == b) for a, b in zip(r_ids, ref_ids)), steps=K,
                   margin_qualified_steps=len(qual), ids_equal_on_margin_qualified_steps=sum(int(r_ids[i] == ref_ids[i]) for i in qual),
                   top1_margin_error_rms=float(np.sqrt(np.mean([s["pair_err"][0] ** 2 for s in steps]))), sigma_reference_bf16=sig)
        table["bf16" if preset is None else preset] = row
    ok = [p_ for p_ in ("all", "mlp", "llm-mlp") if table[p_]["last_hidden_vs_fp32"]["min_cos"] >= 0.99
          and table[p_]["ids_equal_on_margin_qualified_steps"] == table[p_]["margin_qualified_steps"]]
    table["_preset_for_the_bench_line"] = ok[0] if ok else None
    table["_note"] = ("engine (fp8 preset) vs the reference MODULES' fp32 execution at configs[4]'s geometry, prompt 0 of 3; random seeded weights at the true shapes; "
                      "decode steps run the bf16 weights (fp8 serves products of >= 512 rows).  _preset_for_the_bench_line = the widest preset with last-hidden "
                      "min-cos >= 0.99 AND every margin-qualified id equal to the reference's, or null: then no fp8 preset is reported as more than a speed figure")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(table, open(os.path.join(ROOT, "gpurun_out", "r05_fp8_hires_metrics.json"), "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "steps"}) for k, v in table.items()}, indent=1)[:3000])
    # the bf16 row is the pinned one: same bars as test_vs_reference_modules_at_full_depth[hires]
    check(dict(x=table["bf16"]["region_tokens_vs_fp32"]), "x", "region_tokens")
    check(dict(x=table["bf16"]["last_hidden_vs_fp32"]), "x", "llm_final_last_row", slack=2.0)
    assert table["bf16"]["ids_equal_on_margin_qualified_steps"] == table["bf16"]["margin_qualified_steps"]
    # the fp8 rows are BOUNDED, not pinned: first-measurement bars with margin (random weights accumulate ~3 % rms per e4m3 product over 68 blocks)
    for p_ in ("llm-mlp", "mlp", "all"):
        assert table[p_]["region_tokens_vs_fp32"]["min_cos"] >= 0.995 and table[p_]["last_hidden_vs_fp32"]["min_cos"] >= 0.85, (p_, table[p_])


def test_logits_first_token(world, product_library):
    """Prefill logits: the first greedy token equals the oracle's whenever its margin qualifies (the whole-vocabulary maximum deviation
    is recorded in the metrics file; with 152k entries it is an extreme-value statistic that bounds nothing about the argmax)."""
    sig = noise()
    for name in CASES:
        M = run_case(world, name)
        for nm in ("isolated", "composed"):
            m = M[f"logits_{nm}"]
            if m["margin"] > 3.0 * sig:
                assert m["argmax_equal"], f"{name}: first greedy token differs from the oracle's ({nm}) at margin {m['margin']:.3g}"
