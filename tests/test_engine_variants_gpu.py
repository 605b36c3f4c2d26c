"""Engine-level parity for every checkpoint configuration `FO1HFConfig.engine_config()` accepts: the wiring of
`FO1Engine.encode_regions` / `_regions_batch` / `prefill_batch` (vlm_fo1_amd/model.py), not just the kernels.

    region branch   {SimpleFPN 5888, no FPN 8960 (4 captured ViT maps), aux-only 3840, vt-only 2048}
  x projectors      {mlp2x_gelu, linear}                         (multimodal_projector/builder.py:39-115)
  x boxes           {fixture boxes, none -> the dummy box [0,10,0,10]}   (omchat_qwen2_5_vl.py:90-91)
  x aux size        {dynamic (aux = the image, 480x640), squash 768x768 under a 640x480 primary image: the vt box scale differs
                     per axis (omchat_qwen2_5_vl.py:94-99)}
  + region LayerNorm, 'hybrid' position embedding, 'concat_aux_pos' through the engine.

Reference chain: encode_images -> encode_regions -> splice -> Qwen2_5_VLModel.forward (omchat_qwen2_5_vl.py:44-128,135-463), here
the composed CPU oracle (tests/composed_oracle.py) at true channel widths and reduced depth (ViT 4 blocks, LLM 2 layers).
Tolerances: tests/test_e2e_gpu.py's (region / image tokens per-token cosine >= 0.999, max|d| <= 2^-4 max|ref|; logits <= 0.05).
The aux-only rows check the engine's labelled extension (the reference raises UnboundLocalError there, tests/test_oracle_hfre.py)."""
import itertools

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GH, GW = 34, 46            # 640x480 primary image -> 476 x 644 after smart-resize: the metric configuration's patch grid
AUX = {"dynamic": (480, 640), "squash": (768, 768)}
REGION = {"fpn": dict(mm_use_simpleFPN_for_vt=True, mm_region_hidden_size=5888),
          "nofpn": dict(mm_use_simpleFPN_for_vt=False, mm_region_hidden_size=8960),
          "auxonly": dict(mm_use_vision_tower_region_feature=False, mm_use_simpleFPN_for_vt=False, mm_region_hidden_size=3840),
          "vtonly": dict(mm_use_vt_region_feature_only=True, mm_use_simpleFPN_for_vt=True, mm_region_hidden_size=2048)}
SEED = 21
_cache = {}


def make_cfg(**kw):
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config
    from vlm_fo1_amd.vit import ViTConfig
    return FO1Config(vit=ViTConfig(depth=4, fullatt_block_indexes=(0, 1, 2, 3)), llm=LLMConfig(num_layers=2, vocab_size=4096, max_seq=2048), **kw)


def inputs(aux_mode, n_boxes=12):
    """Seeded pixel rows, aux image and fixture boxes (CountBench geometry scaled into the aux image)."""
    key = ("in", aux_mode)
    if key not in _cache:
        from hfre_cases import box_fixtures
        H, W = AUX[aux_mode]
        g = torch.Generator().manual_seed(4)
        pix = torch.randn(GH * GW, 1176, generator=g).bfloat16()
        aux = torch.randn(3, H, W, generator=g).bfloat16()
        it = [x for x in box_fixtures()["countbench"] if len(x["bboxes"]) >= n_boxes][0]
        boxes = torch.tensor(it["bboxes"][:n_boxes], dtype=torch.float32) * torch.tensor([W / it["extent"][0], H / it["extent"][1]] * 2)
        _cache[key] = (pix, aux, boxes)
    return _cache[key]


def oracle_towers(sd, cfg, aux_mode):
    """The tower weights do not depend on the variant (random_weights draws vit -> davit -> fpn -> llm -> projectors from one
    seeded stream), so the slow CPU halves are evaluated once per aux size."""
    import composed_oracle as CO
    pix, aux, _ = inputs(aux_mode)
    if "vit" not in _cache:
        _cache["vit"] = CO.vit(sd, pix, GH, GW, cfg.vit)
    if ("davit", aux_mode) not in _cache:
        _cache[("davit", aux_mode)] = CO.davit_maps(sd, aux)
    return _cache["vit"], _cache[("davit", aux_mode)]


def cpu_state_cached(weights):
    """fp32 CPU copies; the (large) tower parts are converted once — same seed, same draw order, same tensors (spot-checked)."""
    import composed_oracle as CO
    if "towers" not in _cache:
        _cache["towers"] = {k: {n: t.float().cpu() for n, t in weights[k].items()} for k in ("vit", "davit", "fpn")}
    tw = _cache["towers"]
    for part, name in (("vit", "merger.mlp.2.bias"), ("davit", "convs.3.proj.bias"), ("fpn", "simfp_4.2.norm.bias")):
        assert torch.equal(weights[part][name].float().cpu(), tw[part][name]), "tower weights changed between variants"
    sd = dict(tw)
    sd.update(CO.cpu_state({k: weights[k] for k in ("llm", "proj")}))
    return sd


def check(got, ref, what, cos_min=0.999, rel_max=2 ** -4):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    cos = F.cosine_similarity(got, ref, dim=-1)
    rel = (got - ref).abs().max() / ref.abs().max()
    assert cos.min() >= cos_min and rel <= rel_max, f"{what}: min cos {cos.min():.6f}, rel {rel:.4g}"


def run_variant(cfg, aux_mode, with_boxes, region_ln=None):
    import composed_oracle as CO
    from vlm_fo1_amd.model import FO1Engine, random_weights, synthetic_prompt
    weights = random_weights(cfg, "cuda", seed=SEED)
    if region_ln is not None:
        weights["proj"].update({"aux_region_norm.weight": region_ln["aux_w"].cuda(), "aux_region_norm.bias": region_ln["aux_b"].cuda(),
                                "vt_region_norm.weight": region_ln["vt_w"].cuda(), "vt_region_norm.bias": region_ln["vt_b"].cuda()})
    eng = FO1Engine(cfg, weights, "cuda")
    pix, aux, boxes = inputs(aux_mode)
    if not with_boxes:
        boxes = None
    n = 0 if boxes is None else boxes.shape[0]
    ids = synthetic_prompt(n, vocab=4096, seed=2)
    sd = cpu_state_cached(weights)
    (o_tok, o_maps), o_aux = oracle_towers(sd, cfg, aux_mode)
    o_img = CO.projector(o_tok, sd["proj"], "mm_projector.", cfg.mm_projector_type)
    o_feat = CO.region_features(sd, cfg, o_aux, o_maps, boxes, GH, GW, AUX[aux_mode], region_ln=region_ln)
    o_reg = CO.region_tokens(sd, cfg, o_feat)
    # ---- engine: encode_regions alone (the path `generate` takes for the dummy box too), then the whole pass ----
    img_tok, feats = eng.encode_images(pix.cuda(), GH, GW)
    check(img_tok, o_img, "image tokens")
    reg = eng.encode_regions(aux.cuda(), None if boxes is None else boxes.cuda(), feats, GH, GW)
    assert reg.shape == (max(n, 1), cfg.llm.hidden_size)
    check(reg, o_reg, "region tokens (encode_regions)")
    out = eng.prefill(ids, pix.cuda(), (GH, GW), aux.cuda(), None if boxes is None else boxes.cuda())
    ref = CO.llm_prefill(sd, cfg, ids, o_img, o_reg if n else None, GH, GW)
    if n:
        assert torch.equal(out["region_tokens"], reg), "the packed pass and encode_regions must run the same kernels on the same data"
    else:
        assert out["region_tokens"] is None          # no <regionfeat> in the prompt: the dummy features never reach the splice
    assert torch.equal(ref["pos"], out["position_ids"]) and ref["delta"] == out["rope_delta"]
    check(out["last_hidden"], ref["final"][-1:], "final hidden")
    err = (out["logits"].float().cpu() - ref["logits"]).abs().max()
    assert err <= 0.05, f"logits max err {err:.4g}"
    top2 = ref["logits"][0].topk(2).values
    if top2[0] - top2[1] > 0.1:
        assert int(out["next_token"].item()) == int(ref["logits"].argmax())
    return eng, out


@pytest.mark.parametrize("region,proj,with_boxes,aux_mode",
                         list(itertools.product(REGION, ("mlp2x_gelu", "linear"), (True, False), AUX)))
def test_engine_variant(region, proj, with_boxes, aux_mode, product_library):
    cfg = make_cfg(mm_projector_type=proj, mm_projector_aux_type=proj, **REGION[region])
    eng, _ = run_variant(cfg, aux_mode, with_boxes)
    assert eng.capture == {"fpn": "last", "nofpn": "all", "auxonly": "none", "vtonly": "last"}[region]
    assert (eng.fpn is not None) == (region in ("fpn", "vtonly"))


@pytest.mark.parametrize("extra", [dict(mm_apply_region_layer_norm=True), dict(mm_pos_embedding_strategy="hybrid"),
                                   dict(mm_pos_embedding_strategy="feature_map_based"), dict(mm_region_feature_combination="concat_aux_pos"),
                                   dict(mm_apply_position_embedding=False)],
                         ids=lambda d: next(iter(d)) + "=" + str(next(iter(d.values()))))
@pytest.mark.parametrize("region", ["fpn", "nofpn", "auxonly"])
def test_engine_hfre_options(region, extra, product_library):
    """Region LayerNorm (:365-372), feature-map / hybrid position embedding (:327-335), the aux-box embedding ('concat_aux_pos') and no
    embedding at all, through FO1Engine on the three region layouts."""
    if region == "auxonly" and "mm_region_feature_combination" in extra:
        pytest.skip("aux-only already embeds the aux boxes")
    cfg = make_cfg(**REGION[region], **extra)
    ln = None
    if extra.get("mm_apply_region_layer_norm"):
        g = torch.Generator().manual_seed(123)
        ca = 3840
        cv = cfg.mm_region_hidden_size - ca
        ln = dict(aux_w=1 + 0.2 * torch.randn(ca, generator=g), aux_b=0.1 * torch.randn(ca, generator=g),
                  vt_w=1 + 0.2 * torch.randn(max(cv, 1), generator=g), vt_b=0.1 * torch.randn(max(cv, 1), generator=g))
    run_variant(cfg, "dynamic", True, region_ln=ln)


def test_batched_pass_of_a_variant_equals_one_image_passes(ab_library):
    """No-FPN and aux-only through `prefill_batch` with 3 same-geometry images (the stacked-maps HFRE launch of `_regions_batch`):
    every request's region / image tokens equal the one-image pass bit for bit."""
    from vlm_fo1_amd import lib as L
    from vlm_fo1_amd.model import FO1Engine, random_weights, synthetic_prompt
    for region in ("nofpn", "auxonly", "vtonly"):
        cfg = make_cfg(**REGION[region])
        eng = FO1Engine(cfg, random_weights(cfg, "cuda", seed=SEED), "cuda")
        reqs = []
        for i, n in enumerate((9, 12, 4)):
            g = torch.Generator().manual_seed(50 + i)
            pix = torch.randn(GH * GW, 1176, generator=g).bfloat16().cuda()
            aux = torch.randn(3, 480, 640, generator=g).bfloat16().cuda()
            boxes = inputs("dynamic")[2][:n].cuda()
            reqs.append(dict(ids=synthetic_prompt(n, vocab=4096, seed=i), pix=pix, grid=(GH, GW), aux=aux, boxes=boxes))
        try:
            L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")     # pinned tile, no split-K, no GEMV: a row's sum order is M-independent
            L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
            L.check(L.load().fo1_gemm_set_gemv(0), "gemv")
            single = [eng.prefill_batch([r])[0] for r in reqs]
            single = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in single]
            batch = eng.prefill_batch(reqs)
        finally:
            L.load().fo1_gemm_set_variant(0, 0)
            L.load().fo1_gemm_set_splitk(0)
            L.load().fo1_gemm_set_gemv(1)
        for s, b in zip(single, batch):
            for k in ("region_tokens", "image_tokens", "last_hidden", "logits", "next_token"):
                assert torch.equal(s[k], b[k]), f"{region}: {k} differs between the packed pass and the one-image pass"
