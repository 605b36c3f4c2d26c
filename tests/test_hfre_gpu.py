"""GPU parity tests for fo1_hfre_region_pool (through the C-ABI) against the oracle and
the committed golden vectors.

Tolerance (stated once, used everywhere here): fp32 output vs the fp32 direct-algorithm
oracle  rtol 1e-4, atol 2e-5 — the kernel uses the exact separable reformulation, which
re-associates fp32 sums (SURVEY §7 'Hard parts')."""
import os

import numpy as np
import pytest
import torch

from hfre_cases import CASES, checksum, make_case, pyramid_sizes, box_fixtures
from oracle import hfre_oracle as O

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 2e-5
FPN_STRIDES = [3.5, 7, 14, 28]
HERE = os.path.dirname(os.path.abspath(__file__))


def to_dev(case):
    d = dict(case)
    for k in ("aux_maps", "fpn_maps", "vt_maps"):
        if k in case:
            # keep the token-major memory layout of the views
            d[k] = [m.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2) for m in case[k]]
    d["boxes"] = case["boxes"].cuda()
    d["vt_boxes"] = case["vt_boxes"].cuda()
    return d


def engine_out(case_dev, use_vt_boxes=True, worklist=None, **variant):
    from vlm_fo1_amd.hfre import HFREModule
    fpn = case_dev["fpn"]
    gh, gw = case_dev["grid_hw"]
    kw = dict(roi_output_size=7, region_feature_dim=case_dev["region_dim"], apply_position_embedding=True,
              use_vision_tower_region_feature=True, region_feature_combination="concat",
              vision_tower_region_feature_dim=2048 if fpn else 5120, use_simpleFPN_for_vt=fpn,
              simple_fpn=(lambda x: case_dev["fpn_maps"]) if fpn else None)
    ln = variant.pop("ln", None)
    kw.update(variant)
    m = HFREModule(**kw)
    if worklist is not None:
        m.worklist = worklist
    if ln is not None:
        m.set_region_norm(*[ln[k].cuda() for k in ("aux_w", "aux_b", "vt_w", "vt_b")])
    vt_in = torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda") if fpn else case_dev["vt_maps"]
    if use_vt_boxes:
        return m(case_dev["aux_maps"], [case_dev["boxes"]], vt_in, [case_dev["vt_boxes"]]).squeeze(0)
    return m(case_dev["aux_maps"], [case_dev["boxes"]], vt_in, None, vt_scale=case_dev["vt_scale"]).squeeze(0)


def oracle_out(case):
    if case["fpn"]:
        return O.hfre_oracle(case["aux_maps"], case["boxes"], case["fpn_maps"], case["vt_boxes"],
                             region_dim=case["region_dim"], grid_hw=case["grid_hw"], vt_strides=FPN_STRIDES)[0]
    return O.hfre_oracle(case["aux_maps"], case["boxes"], case["vt_maps"], case["vt_boxes"],
                         region_dim=case["region_dim"], grid_hw=case["grid_hw"])[0]


@pytest.mark.parametrize("name", list(CASES))
def test_hfre_vs_golden_and_oracle(name, product_library):
    case = make_case(name)
    g = np.load(os.path.join(HERE, "golden", f"hfre_{name}.npz"))
    assert str(g["checksum"]) == checksum(case)
    got = engine_out(to_dev(case)).cpu()
    assert got.dtype == torch.float32 and got.shape == (case["boxes"].shape[0], case["region_dim"])
    ch = torch.from_numpy(g["channels"]).long()
    torch.testing.assert_close(got[:, ch], torch.from_numpy(g["out"]), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(got, oracle_out(case), rtol=RTOL, atol=ATOL)


def test_hfre_in_kernel_vt_scaling_matches_explicit_vt_boxes():
    case = make_case("demo_fpn")
    d = to_dev(case)
    a = engine_out(d, use_vt_boxes=True)
    b = engine_out(d, use_vt_boxes=False)
    # vt = aux*scale is one fp32 multiply either way; python-float scale vs fp32 tensor scale may
    # differ by an ulp in the box, so compare at the parity tolerance, not bitwise
    torch.testing.assert_close(a, b, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("budget", [16, 100, 4096])
def test_hfre_slice_budget_invariance(budget, ab_library):
    """Row-slicing is a pure work partition: any pixel budget must give the same result
    (to fp32 re-association)."""
    from vlm_fo1_amd import lib as L
    case = make_case("edge_fpn")
    d = to_dev(case)
    ref = engine_out(d).cpu()
    try:
        L.check(L.load().fo1_hfre_set_pixel_budget(budget), "set budget")
        got = engine_out(d).cpu()
    finally:
        L.load().fo1_hfre_set_pixel_budget(0)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(got, oracle_out(case), rtol=RTOL, atol=ATOL)


def test_hfre_deterministic():
    d = to_dev(make_case("countbench30_fpn"))
    a = engine_out(d)
    b = engine_out(d)
    assert torch.equal(a, b), "no atomics: repeated runs must be bit-identical"


def test_hfre_inputs_untouched():
    """Inputs are const (the reference mutates caller box tensors in place, :319,456-463;
    we must not)."""
    d = to_dev(make_case("demo_fpn"))
    b0, v0 = d["boxes"].clone(), d["vt_boxes"].clone()
    engine_out(d)
    assert torch.equal(d["boxes"], b0) and torch.equal(d["vt_boxes"], v0)


def _full_size_case(H, W, n_boxes, seed):
    g = torch.Generator().manual_seed(seed)
    sizes = pyramid_sizes(H, W)
    aux = [torch.randn(h * w, c, generator=g).bfloat16().reshape(h, w, c).permute(2, 0, 1).unsqueeze(0)
           for (h, w), c in zip(sizes, (256, 512, 1024, 2048))]
    gh, gw = round(H / 28) * 2, round(W / 28) * 2
    fpn = []
    for f in (4, 2, 1, 0.5):
        h, w = int(gh * f), int(gw * f)
        fpn.append(torch.randn(h * w, 512, generator=g).bfloat16().reshape(h, w, 512).permute(2, 0, 1).unsqueeze(0))
    it = [x for x in box_fixtures()["pixmo"] if len(x["bboxes"]) == 100][0]
    b = torch.tensor(it["bboxes"], dtype=torch.float32)[:n_boxes]
    ex, ey = it["extent"]
    b = b * torch.tensor([W / ex, H / ey, W / ex, H / ey])
    sw, sh = gw * 14 / W, gh * 14 / H
    return dict(aux_maps=aux, fpn_maps=fpn, fpn=True, grid_hw=(gh, gw), boxes=b, vt_scale=(sw, sh),
                vt_boxes=b * torch.tensor([sw, sh, sw, sh]), region_dim=5888)


def test_hfre_full_size_coco_like_100_boxes():
    """BASELINE config 3 geometry: 640x480 image, 100 real UPN boxes, true channel counts."""
    case = _full_size_case(480, 640, 100, 77)
    got = engine_out(to_dev(case)).cpu()
    torch.testing.assert_close(got, oracle_out(case), rtol=RTOL, atol=ATOL)


def test_hfre_linearity_at_max_size():
    """Size-independent property at the largest supported geometry (1344x1344 -> aux 336..42,
    FPN 384..48): pooling is linear in the maps, so pool(a) + pool(b) == pool(a+b) when a+b is
    exactly representable (b = a here: 2a is exact in bf16)."""
    case = _full_size_case(1344, 1344, 100, 5)
    d = to_dev(case)
    from vlm_fo1_amd.hfre import HFREModule
    one = engine_out(d)
    d2 = dict(d)
    for k in ("aux_maps", "fpn_maps"):
        d2[k] = [(m.permute(0, 2, 3, 1) * 2).contiguous().permute(0, 3, 1, 2) for m in d[k]]
    two = engine_out(d2)
    m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=False,
                   use_vision_tower_region_feature=True, vision_tower_region_feature_dim=2048,
                   use_simpleFPN_for_vt=True, simple_fpn=lambda x: d["fpn_maps"])
    gh, gw = d["grid_hw"]
    nopos = m(d["aux_maps"], [d["boxes"]], torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda"),
              [d["vt_boxes"]]).squeeze(0)
    pos = one - nopos
    torch.testing.assert_close(two - pos, 2 * nopos, rtol=1e-5, atol=1e-5)
    # spot-check 5 boxes against the oracle at this size (the full check would take minutes on CPU)
    sub = dict(case)
    sub["boxes"] = case["boxes"][:5]
    sub["vt_boxes"] = case["vt_boxes"][:5]
    torch.testing.assert_close(one[:5].cpu(), oracle_out(sub), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name", list(CASES))
def test_hfre_worklist_equals_worst_case_grid(name, ab_library):
    """The work-list kernels do the same fp32 operations in the same order as the round-1 worst-case-grid kernels when both slice
    at the same pixel budget: bit-identical.  (The default budgets differ — the work-list form uses one budget for every box
    count — so the budget is pinned here.)"""
    from vlm_fo1_amd import lib as L
    d = to_dev(make_case(name))
    try:
        L.check(L.load().fo1_hfre_set_tuning(8, 512, -1, 0), "work-list form")
        L.check(L.load().fo1_hfre_set_pixel_budget(512), "set budget")
        a = engine_out(d, worklist=True)
        b = engine_out(d, worklist=False)
        again = engine_out(d, worklist=True)
    finally:
        L.load().fo1_hfre_set_pixel_budget(0)
    assert torch.equal(a, b)
    assert torch.equal(a, again), "the item counter is reset per call; list order does not reach the results"


@pytest.mark.parametrize("unroll,chunk,grid", [(16, 512, 2048), (8, 512, 7), (8, 256, 64), (16, 128, 7)])
def test_hfre_worklist_tuning_invariance(unroll, chunk, grid, ab_library):
    """unroll / grid size repartition the work only: bit-identical results (a grid of 7 makes every workgroup walk many items).
    The chunk width changes how many pixel slots a wave has, i.e. the order of the fp32 pixel sum: equal to re-association."""
    from vlm_fo1_amd import lib as L
    d = to_dev(make_case("countbench30_fpn"))
    try:
        L.check(L.load().fo1_hfre_set_tuning(8, 512, -1, 0), "work-list form")
        ref = engine_out(d, worklist=True)
        L.check(L.load().fo1_hfre_set_tuning(unroll, chunk, 0, grid), "set tuning")
        got = engine_out(d, worklist=True)
        again = engine_out(d, worklist=True)
    finally:
        L.load().fo1_hfre_set_tuning(8, 512, 0, 4096)
    assert torch.equal(got, again)
    if chunk == 512:
        assert torch.equal(got, ref)
    else:
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", list(CASES) + ["full_size", "batched"])
def test_hfre_band_form_equals_work_list_form_and_is_invariant_to_the_row_range(name, ab_library):
    """Round 4's band kernel (a workgroup streams a strip of the map once and feeds every box that touches it; A/B build only: measured
    4x slower than the work-list kernel, profiles/r04_hfre_band_form_in_pipeline_4x_slower.json) against the work-list kernel
    (box-major gather): the same separable weights and fp32 fmaf per (pixel, channel), another summation order ->
    equal to fp32 re-association; rows per work item (8 / 32 / one range per map) only change how a box's rows are split into
    partial slices; and run-to-run the band form is bit-identical (a box's accumulator row is owned by one wave)."""
    from vlm_fo1_amd import lib as L
    from vlm_fo1_amd.hfre import HFREModule
    lib = L.load()

    def run():
        if name == "batched":
            cases = [_full_size_case(480, 640, n, 200 + i) for i, n in enumerate((100, 37, 1))]
            stack = lambda key: [torch.cat([c[key][l].permute(0, 2, 3, 1).contiguous() for c in cases], 0).cuda() for l in range(4)]
            aux, fpn = stack("aux_maps"), stack("fpn_maps")
            gh, gw = cases[0]["grid_hw"]
            m = HFREModule(roi_output_size=7, region_feature_dim=cases[0]["region_dim"], apply_position_embedding=True,
                           use_vision_tower_region_feature=True, vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True,
                           simple_fpn=lambda x: [t[:1].permute(0, 3, 1, 2) for t in fpn])
            boxes = torch.cat([c["boxes"] for c in cases]).cuda()
            vtb = torch.cat([c["vt_boxes"] for c in cases]).cuda()
            bi = torch.cat([torch.full((c["boxes"].shape[0],), i, dtype=torch.int32) for i, c in enumerate(cases)]).cuda()
            return m([t[:1].permute(0, 3, 1, 2) for t in aux], [boxes], torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda"), [vtb],
                     batch=3, box_image=bi).squeeze(0).clone()
        d = to_dev(_full_size_case(480, 640, 100, 77) if name == "full_size" else make_case(name))
        return engine_out(d, worklist=True).clone()

    try:
        L.check(lib.fo1_hfre_set_tuning(8, 512, -1, 0), "work-list form")
        ref = run()
        outs = {}
        for rr in (32, 8, 1024):
            L.check(lib.fo1_hfre_set_tuning(8, 512, -2, rr), "band form")
            outs[rr] = run()
            assert torch.equal(outs[rr], run()), "band form is not run-to-run deterministic"
    finally:
        lib.fo1_hfre_set_tuning(8, 512, -1, 32)
    for rr, got in outs.items():
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6, msg=lambda m: f"band form (rows per item {rr}) vs work-list form: {m}")


def test_hfre_band_form_more_boxes_than_one_pass_holds(ab_library):
    """More than 128 boxes on one image: the band kernel accumulates 128 boxes per pass over the strip (box order), the rest in further
    passes — every box's row must still equal what it gets in a call of its own."""
    from vlm_fo1_amd import lib as L
    L.check(L.load().fo1_hfre_set_tuning(8, 512, -2, 32), "band form")
    try:
        _band_many_boxes()
    finally:
        L.load().fo1_hfre_set_tuning(8, 512, -1, 32)


def _band_many_boxes():
    case = _full_size_case(480, 640, 100, 31)
    d = to_dev(case)
    rep = 3                                             # 300 boxes on one image: three passes
    big = dict(d)
    big["boxes"] = d["boxes"].repeat(rep, 1) + torch.arange(rep, device="cuda").repeat_interleave(100).view(-1, 1) * 0.5
    big["vt_boxes"] = d["vt_boxes"].repeat(rep, 1) + torch.arange(rep, device="cuda").repeat_interleave(100).view(-1, 1) * 0.5 * d["vt_scale"][0]
    all_rows = engine_out(big)
    for k in range(rep):
        part = dict(d)
        part["boxes"], part["vt_boxes"] = big["boxes"][100 * k:100 * (k + 1)], big["vt_boxes"][100 * k:100 * (k + 1)]
        assert torch.equal(engine_out(part), all_rows[100 * k:100 * (k + 1)]), f"boxes {100 * k}..: a box's row depends on what else is in the call"


def test_hfre_worklist_graph_replay_under_load():
    """The work-list counter is reset in-stream by the finish kernel (a hipMemsetAsync node in a captured graph faulted on the second
    replay on ROCm 7.2): a captured call replayed many times next to a busy side stream stays bit-identical."""
    d = to_dev(make_case("countbench30_fpn"))
    ref = engine_out(d, worklist=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = engine_out(d, worklist=True)
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    for _ in range(30):
        with torch.cuda.stream(side):
            for _ in range(3):
                a @ a
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, ref)


def test_hfre_variants_vs_reference_golden_and_oracle(product_library):
    """apply_region_layer_norm / concat_aux_pos / use_vt_region_feature_only against the reference HFREModule's outputs
    (tests/golden/hfre_variants.npz) and the oracle.  LayerNorm divides by the row's std (~0.05 here), which scales the pooling's
    fp32 re-association error by 1/std: rtol 1e-4, atol 2e-4 for that variant."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_variant_golden import ln_params
    case = make_case("demo_fpn")
    g = np.load(os.path.join(HERE, "golden", "hfre_variants.npz"))
    assert str(g["checksum"]) == checksum(case)
    d = to_dev(case)
    got = engine_out(d, apply_region_layer_norm=True, ln=ln_params()).cpu()
    torch.testing.assert_close(got, torch.from_numpy(g["ln"]), rtol=1e-4, atol=2e-4)
    got = engine_out(d, region_feature_combination="concat_aux_pos").cpu()
    torch.testing.assert_close(got, torch.from_numpy(g["aux_pos"]), rtol=RTOL, atol=ATOL)
    d2 = dict(d)
    d2["region_dim"] = 2048
    got = engine_out(d2, use_vt_region_feature_only=True).cpu()
    assert got.shape == (case["boxes"].shape[0], 2048)
    torch.testing.assert_close(got, torch.from_numpy(g["vt_only"]), rtol=RTOL, atol=ATOL)
    # feature-map position embedding (reference :327-335): a bf16 table added to every aux level before pooling; 'hybrid' keeps the box
    # embedding too.  The table is built on the host with the reference's expression, so the bf16 sums are the reference's.
    for name, strategy in (("fm_pos", "feature_map_based"), ("hybrid", "hybrid")):
        got = engine_out(d, pos_embedding_strategy=strategy).cpu()
        torch.testing.assert_close(got, torch.from_numpy(g[name]), rtol=RTOL, atol=ATOL)
    assert not torch.equal(torch.from_numpy(g["fm_pos"]), torch.from_numpy(g["hybrid"]))


def test_hfre_vt_only_with_ln_or_fm_strategy_and_aux_only(product_library):
    """ADVICE r2 + the aux-only route.  vt-only: the reference's branch (:293-317) ignores region LayerNorm and the embedding
    strategy — engine == the reference's own outputs for those configurations (== plain vt-only).  aux-only
    (use_vision_tower_region_feature=False): the reference raises UnboundLocalError (golden `aux_only_error`); the engine's extension
    against the oracle's statement of it (pinned piecewise in tests/test_oracle_hfre.py), with and without region LayerNorm."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_variant_golden import ln_params
    case = make_case("demo_fpn")
    g = np.load(os.path.join(HERE, "golden", "hfre_variants.npz"))
    d = to_dev(case)
    d2 = dict(d)
    d2["region_dim"] = 2048
    got = engine_out(d2, use_vt_region_feature_only=True, apply_region_layer_norm=True, ln=ln_params()).cpu()
    torch.testing.assert_close(got, torch.from_numpy(g["vt_only_ln"]), rtol=RTOL, atol=ATOL)
    got = engine_out(d2, use_vt_region_feature_only=True, pos_embedding_strategy="feature_map_based").cpu()
    torch.testing.assert_close(got, torch.from_numpy(g["vt_only_fm_pos"]), rtol=RTOL, atol=ATOL)
    assert str(g["aux_only_error"]) == "UnboundLocalError"
    from vlm_fo1_amd.hfre import HFREModule
    for ln in (None, ln_params()):
        m = HFREModule(roi_output_size=7, region_feature_dim=3840, apply_position_embedding=True, use_vision_tower_region_feature=False,
                       apply_region_layer_norm=ln is not None)
        if ln is not None:
            m.set_region_norm(ln["aux_w"].cuda(), ln["aux_b"].cuda(), None, None)
        got = m(d["aux_maps"], [d["boxes"]]).squeeze(0).cpu()
        ref = O.hfre_oracle(case["aux_maps"], case["boxes"], None, None, region_dim=3840, aux_only=True, region_ln=ln, grid_hw=case["grid_hw"])[0]
        assert got.shape == (case["boxes"].shape[0], 3840)
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-4 if ln is not None else ATOL)
    # no embedding: exactly the aux block of the reference's no-embedding output
    m = HFREModule(roi_output_size=7, region_feature_dim=3840, apply_position_embedding=False, use_vision_tower_region_feature=False)
    got = m(d["aux_maps"], [d["boxes"]]).squeeze(0).cpu()
    torch.testing.assert_close(got, torch.from_numpy(g["nopos"])[:, :3840], rtol=RTOL, atol=ATOL)


def test_hfre_bf16_second_destination_is_the_rne_cast_of_the_fp32_rows():
    """encode_regions casts the fp32 region features to the tower dtype before mm_projector_aux (omchat_qwen2_5_vl.py:106); the finish
    kernel writes that cast itself (fo1_hfre_opts_t.out_bf16): bit-identical to torch's .to(bfloat16), with and without LayerNorm."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_hfre_variant_golden import ln_params
    from vlm_fo1_amd.hfre import HFREModule
    case = make_case("demo_fpn")
    d = to_dev(case)
    gh, gw = d["grid_hw"]
    for ln in (None, ln_params()):
        m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, use_vision_tower_region_feature=True,
                       vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True, simple_fpn=lambda x: d["fpn_maps"],
                       apply_region_layer_norm=ln is not None)
        if ln is not None:
            m.set_region_norm(*[ln[k].cuda() for k in ("aux_w", "aux_b", "vt_w", "vt_b")])
        n = d["boxes"].shape[0]
        o16 = torch.zeros(n, 5888, dtype=torch.bfloat16, device="cuda")
        o32 = m(d["aux_maps"], [d["boxes"]], torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda"), [d["vt_boxes"]], out_bf16=o16).squeeze(0)
        assert torch.equal(o16, o32.to(torch.bfloat16))


def test_hfre_batched_call_equals_per_image(product_library):
    """batch > 1: the maps of B same-size images stacked, all boxes in one launch with box_image — every box's row is bit-identical
    to the one-image call (same slices, same order)."""
    from vlm_fo1_amd.hfre import HFREModule
    B = 3
    cases = [_full_size_case(480, 640, n, 100 + i) for i, n in enumerate((100, 37, 1))]
    devs = [to_dev(c) for c in cases]
    singles = [engine_out(d, use_vt_boxes=False) for d in devs]

    def stack(key, lvl):
        tm = torch.stack([d[key][lvl].permute(0, 2, 3, 1)[0] for d in devs]).contiguous()      # [B,H,W,C]
        return tm[:1].permute(0, 3, 1, 2)                                                      # image-0 view; storage continues
    aux = [stack("aux_maps", i) for i in range(4)]
    fpn = [stack("fpn_maps", i) for i in range(4)]
    boxes = torch.cat([d["boxes"] for d in devs])
    box_image = torch.cat([torch.full((d["boxes"].shape[0],), i, dtype=torch.int32) for i, d in enumerate(devs)]).cuda()
    gh, gw = devs[0]["grid_hw"]
    m = HFREModule(roi_output_size=7, region_feature_dim=5888, apply_position_embedding=True, use_vision_tower_region_feature=True,
                   vision_tower_region_feature_dim=2048, use_simpleFPN_for_vt=True, simple_fpn=lambda x: fpn)
    got = m(aux, [boxes], torch.zeros(1, 1280, gh, gw, dtype=torch.bfloat16, device="cuda"), None, vt_scale=devs[0]["vt_scale"],
            batch=B, box_image=box_image).squeeze(0)
    assert torch.equal(got, torch.cat(singles))
