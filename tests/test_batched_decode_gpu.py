"""Batched greedy decode (vlm_fo1_amd.llm.BatchDecoder over decode.hip): B sequences per weight stream, bookkeeping on the device.

  * a batch of ragged requests generates, per request, exactly the ids the same path generates for the request alone (per-row
    arithmetic of the batched GEMV / split-KV attention does not depend on who shares the launch) — graph replay and eager alike;
  * every generated id is checked against the CPU oracle's greedy decode (oracle/llm_oracle.greedy_decode teacher-forced on the
    engine's ids): it must be the oracle's argmax whenever the oracle's top-1 margin exceeds 2 x the logit tolerance;
  * the stop rule (reference: HF greedy search + mm_utils.py:137-181): a sequence ends right AFTER its stop id, the others go on;
    max_new_tokens caps every sequence."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def build():
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
    from vlm_fo1_amd.vit import ViTConfig
    cfg = FO1Config(vit=ViTConfig(depth=2, fullatt_block_indexes=(1,)), llm=LLMConfig(num_layers=2, vocab_size=4096, max_seq=1024))
    weights = random_weights(cfg, "cuda", seed=31)
    weights["llm"]["embed_tokens.weight"] = (weights["llm"]["embed_tokens.weight"].float() * 4).bfloat16()   # real top-1 margins
    return cfg, weights, FO1Engine(cfg, weights, "cuda")


def requests():
    from test_batched_prefill_gpu import make_request
    return [make_request(60, 500, 399, 7), make_request(61, 333, 711, 33), make_request(62, 96, 120, 2)]


def oracle_logits(cfg, weights, r, forced):
    """Oracle greedy logits for request r, teacher-forced on `forced` (composed CPU oracles, reduced depth)."""
    from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, vit_oracle as VO
    sd = {k: {n: t.float().cpu() for n, t in v.items()} for k, v in weights.items()}
    gh, gw = r["grid"]
    H, W = r["aux"].shape[-2:]

    def mlp2(x, prefix):
        h = F.gelu(F.linear(x, sd["proj"][prefix + "0.weight"], sd["proj"][prefix + "0.bias"]))
        return F.linear(h, sd["proj"][prefix + "2.weight"], sd["proj"][prefix + "2.bias"])

    tokens, maps = VO.vit_forward(sd["vit"], r["pix"].float().cpu(), gh, gw, depth=cfg.vit.depth, n_heads=16, fullatt=cfg.vit.fullatt_block_indexes)
    img = mlp2(tokens, "mm_projector.")
    fpn = [m.bfloat16() for m in FO.fpn_forward(sd["fpn"], maps[-1].bfloat16().float().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))]
    aux_maps, aux_sizes = DO.davit_forward(sd["davit"], r["aux"].float().cpu().unsqueeze(0))
    aux_nchw = [m.bfloat16().reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0) for m, (h, w) in zip(aux_maps, aux_sizes)]
    boxes = r["boxes"].cpu()
    sw, sh = gw * 14 / W, gh * 14 / H
    feat = HO.hfre_oracle(aux_nchw, boxes, fpn, boxes * torch.tensor([sw, sh, sw, sh]), region_dim=5888, grid_hw=(gh, gw), vt_strides=[3.5, 7, 14, 28])[0]
    reg = mlp2(feat.bfloat16().float(), "mm_projector_aux.")
    emb, nb, na = LO.splice(torch.tensor(r["ids"]), sd["llm"]["embed_tokens.weight"], img, reg)
    pos, delta = LO.rope_index(nb, (gh // 2, gw // 2), na)
    kw = dict(n_layers=cfg.llm.num_layers, n_heads=16, n_kv=2, head_dim=128, eps=1e-6, theta=1e6, sections=(16, 24, 24))
    return LO.greedy_decode(sd["llm"], emb, pos, delta, len(forced), forced=forced, **kw)


def test_batched_decode_vs_alone_vs_oracle_and_stop_rule():
    cfg, weights, eng = build()
    reqs = requests()
    K = 12
    got = {}
    for graph in (False, True):
        got[graph] = eng.generate_batch(reqs, max_new_tokens=K, use_graph=graph)
        assert [len(g) for g in got[graph]] == [K] * 3
    assert got[False] == got[True], "eager and graph-replayed batched decode differ"
    alone = [eng.generate_batch([r], max_new_tokens=K, use_graph=True)[0] for r in reqs]
    assert alone == got[True], "a request decodes differently in a batch than alone"
    tol = 0.05
    for r, ids in zip(reqs, got[True]):
        ref_ids, ref_logits = oracle_logits(cfg, weights, r, ids)
        qualified = 0
        for i, t in enumerate(ids):
            top2 = ref_logits[i].topk(2).values
            assert float(ref_logits[i].max() - ref_logits[i][t]) <= 2 * tol, f"step {i}: engine token {t} is not (near-)optimal for the oracle"
            if float(top2[0] - top2[1]) > 2 * tol:
                qualified += 1
                assert t == ref_ids[i], f"step {i}: engine id {t} != oracle greedy id {ref_ids[i]}"
        assert qualified >= K // 2
    # stop rule: sequence 1 stops right after its 5th token; a shared stop id stops whoever emits it; the rest run to the budget
    stop = got[True][1][4]
    out = eng.generate_batch(reqs, max_new_tokens=K, stop_ids=[stop], use_graph=True)
    for b, ids in enumerate(out):
        full = got[True][b]
        cut = full.index(stop) + 1 if stop in full else K
        assert ids == full[:cut], f"sequence {b}: stop rule gave {ids}, expected {full[:cut]}"
    assert len(out[1]) <= 5
    # budget: max_new_tokens = 3
    assert [g[:3] for g in got[True]] == eng.generate_batch(reqs, max_new_tokens=3, use_graph=True)


@pytest.mark.parametrize("rows_per_lane", [0, 1])
def test_gemv_batch_matches_reference(rows_per_lane):
    """fo1_gemv_batch_bf16 (plain / SwiGLU epilogues, fused RMSNorm, K pieces for deep K at M = 8) against torch fp32, with the
    rows-per-lane blocking by M (0, default) and with one row per lane (1)."""
    from test_ops_gpu import gemm_ref, rb
    from vlm_fo1_amd import lib as L, ops
    L.check(L.load().fo1_gemv_batch_set_rows_per_lane(rows_per_lane), "set_rows_per_lane")
    try:
        _gemv_batch_cases(gemm_ref, rb, ops)
    finally:
        L.load().fo1_gemv_batch_set_rows_per_lane(0)


def test_gemv_batch_rows_independent_of_batch():
    """Sequence m's outputs are the same numbers whether it runs alone or with 7 others, and whatever the rows-per-lane blocking:
    the per-(row, sequence) fp32 sum order is fixed by the shape alone (K segments, K split over waves, 8 lanes per row) — what
    lets a request decode identically alone and in a batch."""
    from vlm_fo1_amd import lib as L, ops
    BF = torch.bfloat16
    torch.manual_seed(9)
    for (N, K) in [(2048, 2048), (2048, 11008), (22016 // 4, 2048)]:
        x = (torch.randn(8, K) * 0.5).to(BF).cuda()
        w = (torch.randn(N, K) * 0.05).to(BF).cuda()
        full = ops.gemv_batch(x, w)
        for m in (0, 3, 7):
            assert torch.equal(ops.gemv_batch(x[m:m + 1].contiguous(), w)[0], full[m]), (N, K, m)
        assert torch.equal(ops.gemv_batch(x[:3].contiguous(), w), full[:3])
        assert torch.equal(ops.gemv_batch(x[:4].contiguous(), w), full[:4])
        L.check(L.load().fo1_gemv_batch_set_rows_per_lane(1), "set_rows_per_lane")
        try:
            assert torch.equal(ops.gemv_batch(x, w), full)
        finally:
            L.load().fo1_gemv_batch_set_rows_per_lane(0)


def _gemv_batch_cases(gemm_ref, rb, ops):
    BF = torch.bfloat16
    torch.manual_seed(5)
    for (M, N, K, hb, hr) in [(1, 2048, 2048, False, True), (3, 2560, 2048, True, False), (8, 2048, 11008, False, True), (5, 1000, 264, True, True),
                              (8, 151936 // 8, 2048, False, False)]:
        x = (torch.randn(M, K) * 0.5).to(BF).cuda()
        w = (torch.randn(N, K) * 0.05).to(BF).cuda()
        bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
        res = torch.randn(M, N).to(BF).cuda() if hr else None
        got = ops.gemv_batch(x, w, bias, res)
        ref = gemm_ref(x, w, bias, res, 0)
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item() + 1e-3, f"gemv_batch {M}x{N}x{K}: max err {err:.4g}"
    M, K, Fh = 8, 2048, 11008
    x = (torch.randn(M, K)).to(BF).cuda()
    nw = (1 + 0.1 * torch.randn(K)).to(BF).cuda()
    wg, wu = (torch.randn(Fh, K) * 0.05).to(BF), (torch.randn(Fh, K) * 0.05).to(BF)
    got = ops.gemv_batch(x, ops.interleave_gate_up(wg, wu).cuda(), mode=ops.GB_SWIGLU, norm_weight=nw, norm_eps=1e-6)
    xf = x.float().cpu()
    xn = rb(rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)) * nw.float().cpu())
    g, u = rb(xn @ wg.float().t()), rb(xn @ wu.float().t())
    ref = rb(rb(F.silu(g)) * u)
    err = (got.float().cpu() - ref).abs().max().item()
    assert got.shape == (M, Fh) and err <= 2e-2 * ref.abs().max().item() + 1e-3, f"gemv_batch swiglu+norm: max err {err:.4g}"
