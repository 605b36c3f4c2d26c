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


def test_batched_decode_twelve_sequences_match_alone(ab_library):
    """More than 8 sequences per weight stream (the MFMA kernel carries up to 16 as MFMA columns): 12 ragged requests generate, per
    request, exactly the ids the same path generates for the request alone.  The prefill GEMMs are pinned to one tile (as in
    test_ragged_batch_equals_sequential_bitwise_with_pinned_tile) so that the KV rows a request starts from are the same bits in
    both runs: what is compared is the decode path's independence of the batch."""
    from test_batched_prefill_gpu import make_request
    from vlm_fo1_amd import lib as L
    cfg, weights, eng = build()
    reqs = [make_request(100 + i, 96 + 28 * (i % 4), 120 + 28 * (i % 3), 1 + (5 * i) % 9) for i in range(12)]
    K = 6
    try:
        L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")
        L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
        L.check(L.load().fo1_gemm_set_gemv(0), "gemv")
        got = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)
        assert [len(g) for g in got] == [K] * 12
        for i in (0, 5, 11):
            assert eng.generate_batch([reqs[i]], max_new_tokens=K, use_graph=True)[0] == got[i], f"request {i} decodes differently in a batch of 12"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_gemv(1)


def test_twenty_five_requests_one_decode_group_of_two_mfma_column_groups(ab_library):
    """17..32 sequences ride as TWO 16-column groups per weight fragment (decode_mfma.hip, MM = 32: x fragments in registers for the
    K = 2048 projections, 16-k-step pieces for `down`): 25 ragged requests — one packed prefill pass, ONE decode group — generate, per
    request, exactly the ids the request generates alone (prefill tile pinned, as above), and the stop rule applies per sequence."""
    from test_batched_prefill_gpu import make_request
    from vlm_fo1_amd import lib as L
    cfg, weights, eng = build()
    assert eng.DECODE_MAX_GROUP >= 25
    reqs = [make_request(300 + i, 96 + 28 * (i % 4), 120 + 28 * (i % 3), 1 + (5 * i) % 9) for i in range(25)]
    K = 6
    try:
        L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")
        L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
        L.check(L.load().fo1_gemm_set_gemv(0), "gemv")
        got = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)
        assert [len(g) for g in got] == [K] * 25
        for i in (0, 15, 16, 17, 24):
            assert eng.generate_batch([reqs[i]], max_new_tokens=K, use_graph=True)[0] == got[i], f"request {i} decodes differently in a batch of 25"
        stop = got[21][2]
        out = eng.generate_batch(reqs, max_new_tokens=K, stop_ids=[stop], use_graph=True)
        for b, ids in enumerate(out):
            cut = got[b].index(stop) + 1 if stop in got[b] else K
            assert ids == got[b][:cut], f"sequence {b}: stop rule gave {ids}, expected {got[b][:cut]}"
        # the same 25 in round 2's groups of at most 16 (13 + 12 on two streams): the same ids
        eng.DECODE_MAX_GROUP = 16
        assert eng.generate_batch(reqs, max_new_tokens=K, use_graph=True) == got
    finally:
        eng.DECODE_MAX_GROUP = type(eng).DECODE_MAX_GROUP
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_gemv(1)


def test_twenty_requests_one_prefill_pass_two_decode_groups(ab_library):
    """More requests than a decode group carries (DECODE_MAX_GROUP = 16 here: round 2's one-column-group decode): generate_batch runs ONE
    packed prefill pass over all 20, then decodes them in balanced groups out of the same prefill cache (the later group's prompt K / V
    must survive the first group's decode).  Per request the ids equal the request decoded alone (prefill tile pinned, as above); the
    stop rule applies per sequence in both groups."""
    from test_batched_prefill_gpu import make_request
    from vlm_fo1_amd import lib as L
    cfg, weights, eng = build()
    eng.DECODE_MAX_GROUP = 16
    reqs = [make_request(200 + i, 96 + 28 * (i % 4), 120 + 28 * (i % 3), 1 + (5 * i) % 9) for i in range(20)]
    K = 6
    try:
        L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")
        L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
        L.check(L.load().fo1_gemm_set_gemv(0), "gemv")
        got = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)
        assert [len(g) for g in got] == [K] * 20
        for i in (0, 15, 16, 19):
            assert eng.generate_batch([reqs[i]], max_new_tokens=K, use_graph=True)[0] == got[i], f"request {i} decodes differently in a batch of 20"
        stop = got[17][2]
        out = eng.generate_batch(reqs, max_new_tokens=K, stop_ids=[stop], use_graph=True)
        for b, ids in enumerate(out):
            cut = got[b].index(stop) + 1 if stop in got[b] else K
            assert ids == got[b][:cut], f"sequence {b}: stop rule gave {ids}, expected {got[b][:cut]}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_gemv(1)


@pytest.mark.parametrize("impl,rows_per_lane", [(1, 0), (0, 0), (0, 1)])
def test_gemv_batch_matches_reference(impl, rows_per_lane, ab_library):
    """fo1_gemv_batch_bf16 (plain / SwiGLU epilogues, fused RMSNorm, K pieces for deep K) against torch fp32: the MFMA skinny GEMM
    (impl 1, default; also at M = 16) and the v_dot2 kernel (impl 0) with the rows-per-lane blocking by M (0) and one row per lane (1)."""
    from test_ops_gpu import gemm_ref, rb
    from vlm_fo1_amd import lib as L, ops
    L.check(L.load().fo1_gemv_batch_set_impl(impl), "set_impl")
    L.check(L.load().fo1_gemv_batch_set_rows_per_lane(rows_per_lane), "set_rows_per_lane")
    try:
        _gemv_batch_cases(gemm_ref, rb, ops, m16=impl == 1)
    finally:
        L.load().fo1_gemv_batch_set_rows_per_lane(0)
        L.load().fo1_gemv_batch_set_impl(1)


def _qkv_case(ops, M, seed=3):
    """Inputs of one fused-QKV decode projection (Qwen2.5-VL-3B head geometry) + fresh caches."""
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    H, KV, HD, K, rows = 16, 2, 128, 2048, 512
    x = (torch.randn(M, K, generator=g)).to(BF).cuda()
    w = (torch.randn((H + 2 * KV) * HD, K, generator=g) * 0.05).to(BF).cuda()
    b = (torch.randn((H + 2 * KV) * HD, generator=g) * 0.1).to(BF).cuda()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).cuda()
    ang = torch.rand(rows, HD, generator=g) * 6.28
    cos, sin = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    state = torch.zeros(M, 8, dtype=torch.int32)
    for m in range(M):
        state[m, 0] = 5 + 13 * m          # cache row
        state[m, 1] = 300 - 7 * m         # rope-table row (M <= 32: rows 83 .. 300; cache rows 5 .. 408)
    return dict(x=x, w=w, b=b, nw=nw, cos=cos, sin=sin, state=state.cuda(), H=H, KV=KV, HD=HD, rows=rows)


def _run_qkv(ops, c):
    kc = torch.zeros(c["KV"], c["rows"], c["HD"], dtype=torch.bfloat16, device="cuda")
    vt = torch.zeros(c["KV"] * c["HD"], c["rows"], dtype=torch.bfloat16, device="cuda")
    q = ops.gemv_batch(c["x"], c["w"], c["b"], mode=ops.GB_QKV, norm_weight=c["nw"], norm_eps=1e-6,
                       qkv=dict(n_q=c["H"], n_kv=c["KV"], cos=c["cos"], sin=c["sin"], state=c["state"], kcache=kc, vtcache=vt))
    torch.cuda.synchronize()
    return q.float().cpu(), kc.float().cpu(), vt.float().cpu()


@pytest.mark.parametrize("M", [1, 5, 8, 16, 25, 32])
def test_gemv_mfma_qkv_matches_reference_and_dot2(M, ab_library):
    """Fused QKV epilogue of the MFMA kernel (RMSNorm -> QKV + bias -> bf16 -> mRoPE -> q rows / K row / V^T column at state.pos)
    against a torch restatement of the reference's rounding points (modeling_qwen2_5_vl.py:126-140, 643-685), and — for M <= 8 —
    against the v_dot2 kernel (same rounding points, different fp32 summation order)."""
    from test_ops_gpu import rb
    from vlm_fo1_amd import lib as L, ops
    c = _qkv_case(ops, M)
    q1, k1, v1 = _run_qkv(ops, c)
    H, KV, HD = c["H"], c["KV"], c["HD"]
    xf = c["x"].float().cpu()
    xn = rb(rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)) * c["nw"].float().cpu())
    qkv = rb(xn @ c["w"].float().cpu().t() + c["b"].float().cpu())
    st = c["state"].cpu()
    cos, sin = c["cos"].float().cpu(), c["sin"].float().cpu()
    scale = qkv.abs().max().item()
    for m in range(M):
        pos, tr = int(st[m, 0]), int(st[m, 1])
        heads = qkv[m, :(H + KV) * HD].view(H + KV, HD)
        a, b = heads[:, :64], heads[:, 64:]
        ra = rb(a * cos[tr, :64]) + rb(-b * sin[tr, :64])
        rbb = rb(b * cos[tr, 64:]) + rb(a * sin[tr, 64:])
        rot = rb(torch.cat([ra, rbb], -1))
        assert (q1[m].view(H, HD) - rot[:H]).abs().max().item() <= 2e-2 * scale, f"q rows, sequence {m}"
        assert (k1[:, pos] - rot[H:]).abs().max().item() <= 2e-2 * scale, f"K row, sequence {m}"
        assert (v1[:, pos].view(KV, HD) - qkv[m, (H + KV) * HD:].view(KV, HD)).abs().max().item() <= 2e-2 * scale, f"V^T column, sequence {m}"
    written = torch.zeros(c["rows"], dtype=torch.bool)
    written[st[:M, 0].long()] = True
    assert k1[:, ~written].abs().max().item() == 0 and v1[:, ~written].abs().max().item() == 0, "cache rows of other positions touched"
    if M <= 8:
        L.check(L.load().fo1_gemv_batch_set_impl(0), "set_impl")
        try:
            q0, k0, v0 = _run_qkv(ops, c)
        finally:
            L.load().fo1_gemv_batch_set_impl(1)
        for a, b, what in ((q1, q0, "q"), (k1, k0, "K"), (v1, v0, "V^T")):
            assert (a - b).abs().max().item() <= 1e-2 * scale, what
            assert (a == b).float().mean().item() >= 0.97, f"{what}: MFMA and v_dot2 results should agree almost everywhere"


def test_attention_decode_workgroup_kernel_matches_split_kernel_and_reference(ab_library):
    """Decode attention: the 64-key split-KV kernel + combine (impl 0, default) and the one-workgroup-per-(KV head, sequence) kernel
    (impl 1: tiles round-robin over 8 waves, merged in LDS; 1024-key splits + combine beyond 2048 rows) against fp32 softmax(q k^T / sqrt(d)) v,
    ragged batch, slot starts != 0."""
    from vlm_fo1_amd import lib as L, ops
    BF = torch.bfloat16
    H, KV, HD = 16, 2, 128
    g = torch.Generator().manual_seed(11)
    for slot, lens in ((1024, [1, 63, 64, 65, 651, 1024]), (4096, [700, 2049, 4096]), (2048, [1999] * 16)):
        B = len(lens)
        rows = B * slot
        kc = (torch.randn(KV, rows, HD, generator=g)).to(BF).cuda()
        vt = (torch.randn(KV * HD, rows, generator=g)).to(BF).cuda()
        q = (torch.randn(B, H * HD, generator=g)).to(BF).cuda()
        state = torch.zeros(B, 8, dtype=torch.int32)
        for b, n in enumerate(lens):
            state[b, 2] = b * slot
            state[b, 0] = b * slot + n - 1
        state = state.cuda()
        scale = HD ** -0.5
        out = {}
        for impl in (1, 0):
            L.check(L.load().fo1_attention_decode_set_impl(impl), "set_impl")
            try:
                out[impl] = ops.attention_decode_batch(q, kc, vt, state, slot, H, KV, HD, scale).float().cpu()
            finally:
                L.load().fo1_attention_decode_set_impl(0)
        kf, vf, qf = kc.float().cpu(), vt.float().cpu(), q.float().cpu()
        for b, n in enumerate(lens):
            for h in range(H):
                kv = h // (H // KV)
                keys = kf[kv, b * slot:b * slot + n]                        # [n, HD]
                vals = vf[kv * HD:(kv + 1) * HD, b * slot:b * slot + n]     # [HD, n]
                p = torch.softmax(keys @ qf[b, h * HD:(h + 1) * HD] * scale, 0)
                ref = vals @ p
                for impl in (1, 0):
                    err = (out[impl][b, h * HD:(h + 1) * HD] - ref).abs().max().item()
                    assert err <= 2e-2, f"impl {impl} slot {slot} seq {b} (n={n}) head {h}: {err:.4g}"
        assert (out[1] - out[0]).abs().max().item() <= 2e-2


def test_o_projection_with_the_attention_combine_in_its_prologue_equals_combine_then_gemv_bitwise():
    """Round 6 (decode at <= 2 sequences, the reference's own batch-1 loop): fo1_attention_decode_batch_partials_bf16 + fo1_gemv_attn_combine_bf16 —
    the o-projection sums the split-KV partials itself — against fo1_attention_decode_batch_bf16 (split + combine launch) + fo1_gemv_batch_bf16 with
    the same residual: the SAME BITS (one shared combine routine, csrc/decode_common.h), for one and two sequences, ragged contexts from 1 key to a
    full 2048-row slot, a finished sequence (zero attention row), and against the fp32 reference of the attention."""
    from vlm_fo1_amd import ops
    BF = torch.bfloat16
    H, KV, HD, D = 16, 2, 128, 2048
    g = torch.Generator().manual_seed(23)
    wo = (torch.randn(D, H * HD, generator=g) * 0.03).to(BF).cuda()
    for slot, lens, fin in ((2048, [651], [0]), (2048, [1, 2048], [0, 0]), (1024, [64, 65], [0, 0]), (4096, [700, 3000], [0, 1]), (1024, [130], [1])):
        B = len(lens)
        kc = torch.randn(KV, B * slot, HD, generator=g).to(BF).cuda()
        vt = torch.randn(KV * HD, B * slot, generator=g).to(BF).cuda()
        q = torch.randn(B, H * HD, generator=g).to(BF).cuda()
        res = torch.randn(B, D, generator=g).to(BF).cuda()
        state = torch.zeros(B, 8, dtype=torch.int32)
        for b, n in enumerate(lens):
            state[b, 2] = b * slot
            state[b, 0] = b * slot + n - 1
            state[b, 3] = fin[b]
        state = state.cuda()
        scale = HD ** -0.5
        att = ops.attention_decode_batch(q, kc, vt, state, slot, H, KV, HD, scale)
        want = ops.gemv_batch(att, wo, residual=res)
        part, pstride, chunk = ops.attention_decode_batch_partials(q, kc, vt, state, slot, H, KV, HD, scale)
        got = ops.gemv_attn_combine(part, pstride, state, chunk, H, KV, wo, residual=res)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"slot {slot} lens {lens}: fused o-projection differs from combine + gemv (max |d| {(got.float() - want.float()).abs().max().item():.4g})"
        # and the attention rows themselves against fp32 softmax(q k^T / sqrt(d)) v (what both forms fed the projection)
        kf, vf, qf = kc.float().cpu(), vt.float().cpu(), q.float().cpu()
        a = att.float().cpu()
        for b, n in enumerate(lens):
            if fin[b]:
                assert a[b].abs().max().item() == 0.0
                continue
            for h in (0, 7, 15):
                kv = h // (H // KV)
                p = torch.softmax(kf[kv, b * slot:b * slot + n] @ qf[b, h * HD:(h + 1) * HD] * scale, 0)
                ref = vf[kv * HD:(kv + 1) * HD, b * slot:b * slot + n] @ p
                assert (a[b, h * HD:(h + 1) * HD] - ref).abs().max().item() <= 2e-2


def test_batch_decoder_fused_combine_gives_the_ids_of_the_combine_launch():
    """BatchDecoder at one and two sequences with FUSED_COMBINE_MAX = 2 (product) and 0 (always the combine launch): the same generated ids —
    so a sequence still decodes to the same ids alone and in any batch."""
    from test_batched_prefill_gpu import make_request
    from vlm_fo1_amd.llm import BatchDecoder
    cfg, weights, eng = build()
    reqs = [make_request(500 + i, 96 + 28 * i, 120, 3 + i) for i in range(2)]
    K = 8
    try:
        got = {}
        for fused in (2, 0):
            BatchDecoder.FUSED_COMBINE_MAX = fused
            eng._dec = None
            got[fused] = [eng.generate_batch(reqs[:n], max_new_tokens=K, use_graph=ug) for n in (1, 2) for ug in (False, True)]
        assert got[2] == got[0]
        assert all(len(t) == K for run in got[2] for t in run)
    finally:
        BatchDecoder.FUSED_COMBINE_MAX = 2


def test_gemv_batch_rows_independent_of_batch(ab_library):
    """Sequence m's outputs are the same numbers whether it runs alone or with 7 others, and whatever the rows-per-lane blocking:
    the per-(row, sequence) fp32 sum order is fixed by the shape alone (K segments, K split over waves, 8 lanes per row) — what
    lets a request decode identically alone and in a batch."""
    from vlm_fo1_amd import lib as L, ops
    BF = torch.bfloat16
    torch.manual_seed(9)
    for (N, K) in [(2048, 2048), (2048, 11008), (22016 // 4, 2048)]:
        x16 = (torch.randn(16, K) * 0.5).to(BF).cuda()
        x = x16[:8].contiguous()
        w = (torch.randn(N, K) * 0.05).to(BF).cuda()
        full = ops.gemv_batch(x, w)
        for m in (0, 3, 7):
            assert torch.equal(ops.gemv_batch(x[m:m + 1].contiguous(), w)[0], full[m]), (N, K, m)
        assert torch.equal(ops.gemv_batch(x[:3].contiguous(), w), full[:3])
        assert torch.equal(ops.gemv_batch(x[:4].contiguous(), w), full[:4])
        full16 = ops.gemv_batch(x16, w)                      # 16 sequences (MFMA kernel): 16 staged rows, same sums
        assert torch.equal(full16[:8], full), (N, K, "M=16 vs M=8")
        assert torch.equal(ops.gemv_batch(x16[11:12].contiguous(), w)[0], full16[11]), (N, K, "M=1 vs row 11 of 16")
        x32 = torch.cat([x16, (torch.randn(16, K) * 0.5).to(BF).cuda()])
        full32 = ops.gemv_batch(x32, w)                      # 32 sequences: two column groups per weight fragment, same sums
        assert torch.equal(full32[:16], full16), (N, K, "M=32 vs M=16")
        assert torch.equal(ops.gemv_batch(x32[27:28].contiguous(), w)[0], full32[27]), (N, K, "M=1 vs row 27 of 32")
        assert torch.equal(ops.gemv_batch(x32[:21].contiguous(), w), full32[:21]), (N, K, "M=21 vs M=32")
        L.check(L.load().fo1_gemv_batch_set_rows_per_lane(1), "set_rows_per_lane")
        try:
            assert torch.equal(ops.gemv_batch(x, w), full)
        finally:
            L.load().fo1_gemv_batch_set_rows_per_lane(0)


def test_decode_attention_rows_do_not_depend_on_the_batch_size_bitwise():
    """The split-KV decode attention writes 64-key partials whatever the launch: at 17..32 sequences an item walks FOUR tiles and writes each tile's
    partial separately (round 6), below that an item is one tile — a sequence's attention row is the same bits alone, in 16 and in 25."""
    from vlm_fo1_amd import ops
    BF = torch.bfloat16
    H, KV, HD, slot = 16, 2, 128, 2048
    g = torch.Generator().manual_seed(3)
    lens = [1, 63, 64, 65, 129, 255, 256, 257, 651, 700, 1023, 1024, 1025, 1999, 2048, 5, 333, 900, 1500, 77, 640, 641, 12, 2047, 512]
    B = len(lens)
    kc = torch.randn(KV, B * slot, HD, generator=g).to(BF).cuda()
    vt = torch.randn(KV * HD, B * slot, generator=g).to(BF).cuda()
    q = torch.randn(B, H * HD, generator=g).to(BF).cuda()
    state = torch.zeros(B, 8, dtype=torch.int32)
    for b, n in enumerate(lens):
        state[b, 2] = b * slot
        state[b, 0] = b * slot + n - 1
    state[19, 3] = 1                                    # a finished sequence: zero row in every batch
    state = state.cuda()
    scale = HD ** -0.5
    full = ops.attention_decode_batch(q, kc, vt, state, slot, H, KV, HD, scale).clone()
    assert full[19].abs().max().item() == 0.0
    for lo, hi in ((0, 16), (3, 4), (8, 9), (14, 15), (24, 25), (16, 25), (0, 17)):
        got = ops.attention_decode_batch(q[lo:hi].contiguous(), kc, vt, state[lo:hi].contiguous(), slot, H, KV, HD, scale)
        assert torch.equal(got, full[lo:hi]), (lo, hi, (got.float() - full[lo:hi].float()).abs().max().item())
    kf, vf, qf = kc.float().cpu(), vt.float().cpu(), q.float().cpu()
    for b in (0, 3, 8, 14, 24):
        n = lens[b]
        for h in (0, 9):
            kv = h // (H // KV)
            pr = torch.softmax(kf[kv, b * slot:b * slot + n] @ qf[b, h * HD:(h + 1) * HD] * scale, 0)
            ref = vf[kv * HD:(kv + 1) * HD, b * slot:b * slot + n] @ pr
            assert (full[b, h * HD:(h + 1) * HD].float().cpu() - ref).abs().max().item() <= 2e-2


def test_deep_k_eight_row_units_at_17_to_26_sequences_equal_the_sixteen_row_units_bitwise(ab_library):
    """Round 6: `down` (K = 11008) at 17..26 sequences runs on 8-row units (256 workgroups) with the launch's own x rows staged per piece; 27..32
    sequences and the A/B switch keep the 16-row units (128 workgroups).  The same sums: rows of a 32-sequence launch == the same rows launched as
    17 / 25 / 26 sequences == the rows with the 8-row units switched off, with a residual operand and a ragged K."""
    from vlm_fo1_amd import lib as L, ops
    g = torch.Generator().manual_seed(17)
    for N, K in ((2048, 11008), (1280, 4096 + 192)):
        w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
        x32 = torch.randn(32, K, generator=g).bfloat16().cuda()
        r32 = torch.randn(32, N, generator=g).bfloat16().cuda()
        full = ops.gemv_batch(x32, w, residual=r32)                 # 32 sequences: 16-row units
        for M in (17, 21, 25, 26, 27):
            got = ops.gemv_batch(x32[:M].contiguous(), w, residual=r32[:M].contiguous())
            assert torch.equal(got, full[:M]), (N, K, M, (got.float() - full[:M].float()).abs().max().item())
            try:
                L.check(L.load().fo1_gemv_batch_set_impl(5), "16-row units at 9..32 sequences")
                assert torch.equal(ops.gemv_batch(x32[:M].contiguous(), w, residual=r32[:M].contiguous()), got), (N, K, M, "A/B")
            finally:
                L.load().fo1_gemv_batch_set_impl(1)
        ref = (x32.float() @ w.float().t()).bfloat16().float() + r32.float()
        assert (full.float() - ref).abs().max().item() <= 0.05 * ref.abs().max().item() + 0.05


def _gemv_batch_cases(gemm_ref, rb, ops, m16=False):
    BF = torch.bfloat16
    torch.manual_seed(5)
    shapes = [(1, 2048, 2048, False, True), (3, 2560, 2048, True, False), (8, 2048, 11008, False, True), (5, 1000, 264, True, True),
              (8, 151936 // 8, 2048, False, False)]
    if m16:
        shapes += [(16, 2048, 11008, True, True), (13, 151936 // 8, 2048, False, False), (16, 1000, 264, True, True),
                   (32, 2048, 11008, True, True), (25, 151936 // 8, 2048, False, False), (17, 1000, 264, True, True), (32, 2560, 2048, True, True)]
    for (M, N, K, hb, hr) in shapes:
        x = (torch.randn(M, K) * 0.5).to(BF).cuda()
        w = (torch.randn(N, K) * 0.05).to(BF).cuda()
        bias = (torch.randn(N) * 0.1).to(BF).cuda() if hb else None
        res = torch.randn(M, N).to(BF).cuda() if hr else None
        got = ops.gemv_batch(x, w, bias, res)
        ref = gemm_ref(x, w, bias, res, 0)
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item() + 1e-3, f"gemv_batch {M}x{N}x{K}: max err {err:.4g}"
    for M in ((16, 25, 32) if m16 else (8,)):
        K, Fh = 2048, 11008
        x = (torch.randn(M, K)).to(BF).cuda()
        nw = (1 + 0.1 * torch.randn(K)).to(BF).cuda()
        wg, wu = (torch.randn(Fh, K) * 0.05).to(BF), (torch.randn(Fh, K) * 0.05).to(BF)
        got = ops.gemv_batch(x, ops.interleave_gate_up(wg, wu).cuda(), mode=ops.GB_SWIGLU, norm_weight=nw, norm_eps=1e-6)
        xf = x.float().cpu()
        xn = rb(rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)) * nw.float().cpu())
        g, u = rb(xn @ wg.float().t()), rb(xn @ wu.float().t())
        ref = rb(rb(F.silu(g)) * u)
        err = (got.float().cpu() - ref).abs().max().item()
        assert got.shape == (M, Fh) and err <= 2e-2 * ref.abs().max().item() + 1e-3, f"gemv_batch swiglu+norm M={M}: max err {err:.4g}"
