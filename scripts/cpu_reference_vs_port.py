"""BUILD CONTAINER ONLY (needs /root/reference; nothing of it is copied — the modules are imported in place): is bench.py's
`cpu_baseline` (`kind: port`: the oracle's operators, oracle/*.py) a fair stand-in for the reference's OWN modules on a CPU?

bench.py cannot time the reference itself: /root/reference does not exist on the GPU box and nothing the bench runs may read it
(VERDICT r4 missing #7 asks for a `kind: reference` leg).  This script answers the question where the reference does exist: the
`metric` workload of tests/fulldepth_case.py (1 image 640x480 x 100 boxes, the BASELINE metric configuration; full depth: 32 ViT
blocks, DaViT-L, SimpleFPN, HFRE, 36 LLM layers), prefill to the first greedy token, fp32, the SAME host threads for both legs:

  reference   vendored Qwen2_5_VisionTransformer through the reference's custom_forward / VisionFeaturesGather, the reference's DaViT,
              SimpleFP and HFREModule (roi_align = oracle/roi_align_ref.c: torchvision is not installed), mlp2x_gelu projectors, the
              vendored Qwen2_5_VLModel (sdpa attention) — the module set of tests/golden/make_fulldepth_ref.py
  port        the oracle stage functions exactly as bench.cpu_baseline.one_pass calls them

    python scripts/cpu_reference_vs_port.py [out.json] [reps]          # ~10 min on 8 cores

-> profiles/r05_cpu_baseline_reference_vs_port_build_container.json; bench.py quotes its ratio in `cpu_baseline.reference_modules_check`."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import fulldepth_case as FC  # noqa: E402
from make_fulldepth_ref import build_llm  # noqa: E402
from measure_bf16_floor import build_modules  # noqa: E402
from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, reference_loader as R, vit_oracle as VO  # noqa: E402


def median_pass(fn, reps):
    fn()                                             # warm-up
    runs = [fn() for _ in range(reps)]
    totals = sorted(sum(t.values()) for t in runs)
    total = totals[len(totals) // 2]
    t = [r for r in runs if sum(r.values()) == total][0]
    return dict(seconds_per_image=round(total, 2), pass_seconds=[round(v, 2) for v in totals], stage_seconds={k: round(v, 3) for k, v in t.items()})


def main():
    assert R.available(), "/root/reference is needed (build container only)"
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_cpu_baseline_reference_vs_port_build_container.json")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    cfg = FC.full_config()
    W, cks = FC.cpu_weights(cfg)
    case = FC.build_case("metric")
    gh, gw = case["grid"]
    H, Wd = case["img_hw"]
    ids, boxes = case["groups"][0]
    sw, sh = gw * 14 / Wd, gh * 14 / H
    vt_boxes = boxes * torch.tensor([sw, sh, sw, sh])
    head = W["llm"]["lm_head.weight"].float()

    # ---- reference modules ----------------------------------------------------------------------------------------------------------
    mods = build_modules(cfg, W, with_llm=False)
    llm = build_llm(cfg, W).float()
    HFREModule, _, _ = HO.load_reference_hfre()
    hfre = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, pos_embedding_strategy="bbox_based",
                      use_vt_region_feature_only=False, use_vision_tower_region_feature=True, region_feature_combination="concat",
                      apply_region_layer_norm=False, vision_tower_region_feature_dim=2048, vision_tower_spatial_scale=1 / 14,
                      use_simpleFPN_for_vt=True, aux_vision_tower_spatial_scale=0.25, aux_vision_tower_region_feature_dims=[256, 512, 1024, 2048])
    hfre.simple_fpn = mods["fpn"]
    emb_w = W["llm"]["embed_tokens.weight"].float()
    ref_tok = {}

    def reference_pass():
        t = {}
        with torch.no_grad():
            t0 = time.perf_counter()
            vit, enc = mods["vit"], mods["enc"]
            gather = enc.VisionFeaturesGather()
            vit.vision_features_gather = gather
            tokens = enc.custom_forward(vit, case["pix"].float(), torch.tensor([[1, gh, gw]]))
            maps = gather.extract_multi_level_features()[0]
            t["vit"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            aux = mods["davit"].forward_features(case["aux"].float().unsqueeze(0))["image_features"]
            t["davit"] = time.perf_counter() - t0
            t0 = time.perf_counter()                 # the reference's HFREModule runs SimpleFP inside its __call__
            feat = hfre(aux_multi_level_features=aux, vt_multi_level_features=maps[-1], aux_boxes=[boxes.clone()], vt_boxes=[vt_boxes.clone()]).squeeze(0)
            t["fpn+hfre"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            reg, img = mods["proj"]["mm_projector_aux"](feat), mods["proj"]["mm_projector"](tokens)
            t["projectors"] = time.perf_counter() - t0
            emb, nb, na = LO.splice(torch.tensor(ids), emb_w, img, reg)
            pos, _ = LO.rope_index(nb, (gh // 2, gw // 2), na)
            t0 = time.perf_counter()
            o = llm(inputs_embeds=emb[None], position_ids=pos[:, None, :], use_cache=True)
            t["llm_prefill"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            ref_tok["id"] = int((o.last_hidden_state[0, -1:] @ head.t()).argmax())
            t["lm_head"] = time.perf_counter() - t0
        return t

    ref = median_pass(reference_pass, reps)
    print("reference modules:", json.dumps(ref), flush=True)
    del mods, llm, hfre

    # ---- the port (bench.cpu_baseline.one_pass) --------------------------------------------------------------------------------------
    sd = {k: {n: t.float() for n, t in v.items()} for k, v in W.items()}
    kw = dict(n_layers=cfg.llm.num_layers, n_heads=cfg.llm.num_heads, n_kv=cfg.llm.num_kv_heads, head_dim=cfg.llm.head_dim,
              eps=cfg.llm.rms_norm_eps, theta=cfg.llm.rope_theta, sections=cfg.llm.mrope_section)
    pix, aux_img = case["pix"].float(), case["aux"].float().unsqueeze(0)
    port_tok = {}

    def mlp2(x, prefix):
        h = F.gelu(F.linear(x, sd["proj"][prefix + "0.weight"], sd["proj"][prefix + "0.bias"]))
        return F.linear(h, sd["proj"][prefix + "2.weight"], sd["proj"][prefix + "2.bias"])

    def port_pass():
        t = {}
        t0 = time.perf_counter()
        tokens, maps = VO.vit_forward(sd["vit"], pix, gh, gw, depth=cfg.vit.depth, n_heads=cfg.vit.num_heads, fullatt=cfg.vit.fullatt_block_indexes)
        t["vit"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        aux_maps, aux_sizes = DO.davit_forward(sd["davit"], aux_img)
        t["davit"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        fpn = FO.fpn_forward(sd["fpn"], maps[-1].reshape(gh, gw, -1).permute(2, 0, 1).unsqueeze(0))
        aux_nchw = [m.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(aux_maps, aux_sizes)]
        feat = HO.hfre_oracle(aux_nchw, boxes, fpn, vt_boxes, region_dim=cfg.mm_region_hidden_size, grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
        t["fpn+hfre"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        reg, img = mlp2(feat, "mm_projector_aux."), mlp2(tokens, "mm_projector.")
        t["projectors"] = time.perf_counter() - t0
        emb, nb, na = LO.splice(torch.tensor(ids), sd["llm"]["embed_tokens.weight"], img, reg)
        pos, _ = LO.rope_index(nb, (gh // 2, gw // 2), na)
        t0 = time.perf_counter()
        hid, _ = LO.llm_forward_cached(sd["llm"], emb, pos, None, **kw)
        t["llm_prefill"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        port_tok["id"] = int((hid[-1:] @ head.t()).argmax())
        t["lm_head"] = time.perf_counter() - t0
        return t

    port = median_pass(port_pass, reps)
    print("oracle port:", json.dumps(port), flush=True)
    res = dict(note="scripts/cpu_reference_vs_port.py in the BUILD CONTAINER (the only place /root/reference exists): the reference's own modules against "
                    "the oracle port that bench.py's cpu_baseline times on the GPU box, same workload (tests/fulldepth_case.py `metric`: 640x480, 100 boxes, "
                    "full depth, prefill to the first greedy token), same host threads, torch fp32, warm-up 1 + median of %d passes each" % reps,
               host_threads=threads, host_cpus=os.cpu_count(), torch=torch.__version__, weight_checksums=cks,
               reference_modules=ref, oracle_port=port,
               port_over_reference_seconds=round(port["seconds_per_image"] / ref["seconds_per_image"], 3),
               first_token_reference=ref_tok.get("id"), first_token_port=port_tok.get("id"))
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res)[:600])


if __name__ == "__main__":
    main()
