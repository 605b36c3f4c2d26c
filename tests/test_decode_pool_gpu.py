"""The decode pool (vlm_fo1_amd.llm.DecodePool; scheduler vlm_fo1_amd.serving.PoolService): 64 / 128 sequence
slots per weight stream, sequences of different prefill passes sharing every decode step (the loop it replaces is the
reference's one-image-at-a-time `generate`, omchat_qwen2_5_vl.py:143-155 + HF greedy search, stop rule mm_utils.py:137-181).

  * the step's products at M = 64 / 128 rows and fo1_pool_qkv_post_bf16 (mRoPE at each slot's position + cache append) against torch
    fp32 with the reference's rounding points;
  * a sequence's ids do not depend on its slot or its neighbours, eager == graph replay; against the <= 32-sequence BatchDecoder (other
    fp32 sum orders) most sequences agree outright, and every id is checked against the CPU oracle's greedy decode;
  * the stop rule and the budget, slots re-used by later joins while other sequences are mid-flight;
  * the scheduler: passes submitted from several replica threads come back with exactly the ids a direct pool run gives.
Runs on the PRODUCT library (no pins needed)."""
import math
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(x):
    return x.to(BF).float()


def _close(got, ref, what, ulps=1.0, rare=2e-3, mag=None):
    """bf16 results of fp32 sums in a different order: within `ulps` bf16 ulps except a `rare` fraction at a rounding boundary (2 ulps).
    `mag`: the magnitude whose ulp counts when the result went through an EARLIER rounding at a larger value (bf16(sum) + residual:
    a one-ulp flip of the sum is several ulps of a small result)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = ref.abs().max().item()
    tol = ulps * 2.0 ** -7 * (ref.abs() if mag is None else mag.float().cpu()) + 2.0 ** -9 * scale * 0.02
    bad = (got - ref).abs() > tol
    assert bad.float().mean().item() <= rare, f"{what}: {int(bad.sum())}/{bad.numel()} beyond {ulps} bf16 ulps"
    assert ((got - ref).abs() <= 2 * tol + 1e-6).all(), f"{what}: max |d| {float((got - ref).abs().max()):.4g} at scale {scale:.3g}"


@pytest.mark.parametrize("P", [64, 128])
def test_pool_projections_and_qkv_post_match_reference(P, product_library):
    """The pool step's products at M = P rows (fo1_gemm_bf16 on the tiles the dispatcher picks for a weight stream: 128 x 128 rings for
    gate/up and lm_head-sized shapes, split-K for down) and fo1_pool_qkv_post_bf16 against torch fp32 with the reference's rounding points."""
    from vlm_fo1_amd import ops
    g = torch.Generator().manual_seed(11 + P)

    def rnd(*s, sc=1.0):
        return (torch.randn(*s, generator=g) * sc).to(BF).cuda()

    for N, K in ((2048, 2048), (2048, 11008)):          # o / down: bias-free, residual
        x, w, res = rnd(P, K), rnd(N, K, sc=0.03), rnd(P, N)
        y = ops.gemm(x, w, residual=res)
        sm = rb(x.float() @ w.float().t())
        ref = rb(sm + res.float())
        _close(y, ref, f"plain+res {N}x{K}", mag=sm.abs() + ref.abs())
    N, K = 128 * 131, 2048                               # lm_head-like: >= 128 row tiles -> the 128 x 128 ring at 64 < M <= 128
    x, w = rnd(P, K), rnd(N, K, sc=0.03)
    _close(ops.gemm(x, w), rb(x.float() @ w.float().t()), "lm_head-like")
    F_, K = 11008, 2048                                  # SwiGLU over 16-row interleaved gate / up rows at the true width
    gate, up, x = rnd(F_, K, sc=0.03), rnd(F_, K, sc=0.03), rnd(P, K)
    got = ops.gemm(x, ops.interleave_gate_up(gate, up).cuda(), act=ops.ACT_SWIGLU16)
    gt, u = rb(x.float() @ gate.float().t()), rb(x.float() @ up.float().t())
    _close(got, rb(rb(torch.nn.functional.silu(gt)) * u), "swiglu", ulps=2.0, rare=5e-3)
    # the pre-tiled weight copy the pool streams gate/up and lm_head from (fo1_gemm_bf16_wtiled): the same kernel on another address pattern -> same bits
    # (round 5: a measured no-gain form, include/fo1_ab.h — the test / bench build only)
    from vlm_fo1_amd import lib as L
    wgu = ops.interleave_gate_up(gate, up).cuda()
    wl, xb, rl = rnd(128 * 9, K, sc=0.03), rnd(P, K), rnd(P, 128 * 9)
    bl = rnd(128 * 9, sc=0.1)
    with L.use_ab():
        got_t = ops.gemm_wtiled(x, ops.tile_weight(wgu), act=ops.ACT_SWIGLU16)
        ref_t = ops.gemm_wtiled(xb, ops.tile_weight(wl), bl, rl)
        torch.cuda.synchronize()
    if P > 64:
        assert torch.equal(got_t, got), "tiled gate/up (same 128 x 128 ring tile as the row-major call)"
    else:
        _close(got_t, got, "tiled gate/up (64 x 128 tile against the row-major call's 64 x 64)", ulps=1.0)
    _close(ref_t, rb(rb(xb.float() @ wl.float().t() + bl.float()) + rl.float()), "tiled plain + bias + residual", mag=(xb.float() @ wl.float().t()).abs().cpu() + 4)
    # ---- q/k/v: bias -> bf16 (GEMM), then mRoPE at the slot's table row -> q rows in place, K rows, V^T columns at the slot's cache row ----
    H, KV, HD, K, rows = 16, 2, 128, 2048, 1024
    x, w, b = rnd(P, K), rnd((H + 2 * KV) * HD, K, sc=0.05), rnd((H + 2 * KV) * HD, sc=0.1)
    ang = torch.rand(rows, HD, generator=g) * 6.28
    cos, sin = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    st = torch.zeros(P, 8, dtype=torch.int32)
    st[:, 0] = torch.arange(P) * 7 + 3          # distinct cache rows
    st[:, 1] = 900 - 5 * torch.arange(P)        # rope-table rows
    kc = torch.zeros(KV, rows, HD, dtype=BF, device="cuda")
    vt = torch.zeros(KV * HD, rows, dtype=BF, device="cuda")
    qkv16 = ops.gemm(x, w, b)
    qkv = qkv16.float().clone()                  # the bf16 rows the post-processing starts from
    ops.pool_qkv_post(qkv16, H, KV, HD, cos, sin, st.cuda(), kc, vt)
    torch.cuda.synchronize()
    _close(qkv, rb(x.float() @ w.float().t() + b.float()), "qkv product")
    cf, sf = cos.float(), sin.float()
    heads = qkv[:, :(H + KV) * HD].view(P, H + KV, HD)
    tr = st[:, 1].long().cuda()
    a, bb = heads[..., :64], heads[..., 64:]
    c1, s1, c2, s2 = cf[tr, None, :64], sf[tr, None, :64], cf[tr, None, 64:], sf[tr, None, 64:]
    rot = rb(torch.cat([rb(a * c1) + rb(-bb * s1), rb(bb * c2) + rb(a * s2)], -1))
    assert torch.equal(qkv16[:, :H * HD].float().view(P, H, HD), rot[:, :H]), "rotated q rows (exact: same roundings from the same bf16 rows)"
    pos = st[:, 0].long().cuda()
    assert torch.equal(kc.float()[:, pos].permute(1, 0, 2), rot[:, H:]), "K rows"
    assert torch.equal(vt.float()[:, pos].t().reshape(P, KV, HD), qkv[:, (H + KV) * HD:].view(P, KV, HD)), "V^T columns"
    written = torch.zeros(rows, dtype=torch.bool, device="cuda")
    written[pos] = True
    assert kc[:, ~written].abs().max().item() == 0 and vt[:, ~written].abs().max().item() == 0, "cache rows of other positions touched"


@pytest.mark.parametrize("P", [64, 128])
def test_pool_splitk_planes_and_fused_consumers(P, product_library):
    """fo1_gemm_bf16_partials (fp32 split-K planes, no epilogue) and the two kernels that consume them in the pool step:
    fo1_splitk_residual_rmsnorm_bf16 — x = bf16(bf16(sum_z plane_z) + residual) EXACTLY (the same fp32 adds in the same order), the fused
    RMSNorm within one bf16 ulp of torch fp32 — and fo1_pool_qkv_post_partials_bf16 == fo1_pool_qkv_post_bf16 on bf16(sum_z plane_z + bias), bit for bit."""
    from vlm_fo1_amd import ops
    g = torch.Generator().manual_seed(23 + P)

    def rnd(*s, sc=1.0):
        return (torch.randn(*s, generator=g) * sc).to(BF).cuda()

    def planes_of(x, w, splits):
        N = w.shape[0]
        part = torch.full((splits * P * N + 64,), float("nan"), dtype=torch.float32, device="cuda")
        s = ops.gemm_partials(x, w, splits, part)
        torch.cuda.synchronize()
        assert 2 <= s <= splits and torch.isnan(part[s * P * N:]).all(), "planes beyond the effective split count written"
        pl = part[:s * P * N].view(s, P, N)
        tot = pl[0].clone()
        for z in range(1, s):
            tot += pl[z]
        ref = x.float() @ w.float().t()
        assert (tot - ref).abs().max().item() <= 2e-5 * K ** 0.5 * ref.abs().max().item(), "sum of the planes != the product"
        return part, s, tot

    for N, K, splits in ((2048, 2048, 4), (2048, 11008, 8)):          # o / down
        x, w, res, nw = rnd(P, K), rnd(N, K, sc=0.03), rnd(P, N), (1 + 0.1 * torch.randn(N, generator=g)).to(BF).cuda()
        part, s, tot = planes_of(x, w, splits)
        xo, xn = torch.empty(P, N, dtype=BF, device="cuda"), torch.empty(P, N, dtype=BF, device="cuda")
        ops.splitk_residual_rmsnorm(part, s, res, nw, 1e-6, xo, xn)
        torch.cuda.synchronize()
        x_ref = rb(rb(tot) + res.float())
        assert torch.equal(xo.float(), x_ref), f"x_out {N}x{K}"
        rstd = torch.rsqrt((x_ref.double() ** 2).mean(-1, keepdim=True) + 1e-6).float()
        _close(xn, rb(nw.float() * rb(x_ref * rstd)), f"fused RMSNorm {N}x{K}")
        # in place on the residual buffer, as the pool step calls it
        res2 = res.clone()
        ops.splitk_residual_rmsnorm(part, s, res2, nw, 1e-6, res2, xn)
        torch.cuda.synchronize()
        assert torch.equal(res2, xo), "in-place x_out"
    # gate/up against the 16-row interleaved weight at the true width (the 128 x 256 tile, 3 planes) -> fo1_splitk_swiglu_bf16: EXACTLY the
    # SwiGLU of the bf16-rounded plane sums
    F_, K = 11008, 2048
    gate, up, x = rnd(F_, K, sc=0.03), rnd(F_, K, sc=0.03), rnd(P, K)
    part, s, tot = planes_of(x, ops.interleave_gate_up(gate, up).cuda(), 3)
    from vlm_fo1_amd import lib as L
    with L.use_ab():          # (round 5: a measured no-gain form, include/fo1_ab.h — the test / bench build only)
        got = ops.splitk_swiglu(part, s, P, 2 * F_)
        torch.cuda.synchronize()
    t4 = tot.view(P, F_ // 16, 2, 16)
    gt, u = rb(t4[:, :, 0].reshape(P, F_)), rb(t4[:, :, 1].reshape(P, F_))
    ref = rb(rb(gt * torch.sigmoid(gt)) * u)
    _close(got, ref, "splitk swiglu", ulps=1.0, rare=2e-3)          # (silu: the kernel's exp / rcp against torch's sigmoid)
    _close(got, rb(rb(torch.nn.functional.silu(rb(x.float() @ gate.float().t()))) * rb(x.float() @ up.float().t())), "splitk swiglu vs fp32 product", ulps=2.0, rare=5e-3)
    H, KV, HD, K, rows = 16, 2, 128, 2048, 1024
    x, w, b = rnd(P, K), rnd((H + 2 * KV) * HD, K, sc=0.05), rnd((H + 2 * KV) * HD, sc=0.1)
    ang = torch.rand(rows, HD, generator=g) * 6.28
    cos, sin = ang.cos().to(BF).cuda(), ang.sin().to(BF).cuda()
    st = torch.zeros(P, 8, dtype=torch.int32)
    st[:, 0] = torch.arange(P) * 7 + 3
    st[:, 1] = 900 - 5 * torch.arange(P)
    st = st.cuda()
    part, s, tot = planes_of(x, w, 4)
    q = torch.zeros(P, H * HD, dtype=BF, device="cuda")
    kc, vt = torch.zeros(KV, rows, HD, dtype=BF, device="cuda"), torch.zeros(KV * HD, rows, dtype=BF, device="cuda")
    ops.pool_qkv_post_partials(part, s, b, q, H, KV, HD, cos, sin, st, kc, vt)
    qkv_ref = (tot + b.float()).to(BF)
    kc2, vt2 = torch.zeros_like(kc), torch.zeros_like(vt)
    ops.pool_qkv_post(qkv_ref, H, KV, HD, cos, sin, st, kc2, vt2)
    torch.cuda.synchronize()
    assert torch.equal(q, qkv_ref[:, :H * HD]) and torch.equal(kc, kc2) and torch.equal(vt, vt2), "q rows / K rows / V^T columns"


def test_pool_fused_splitk_step_against_unfused(product_library):
    """The pool step with split-K planes + fused consumers (8 launches per layer) against the same step on plain GEMM epilogues + separate
    RMSNorm launches (11): other fp32 sum orders in q/k/v and o, so ids may differ at near-ties only."""
    from vlm_fo1_amd.llm import DecodePool
    cfg, weights, eng = _engine()
    reqs = _requests(9)
    eng.prefill_batch(reqs, use_graph=False)
    hp, first = eng._last_batch, eng._last_next_tokens.clone()
    K = 12
    pool = DecodePool(eng.llm, slots=64, slot_rows=512)
    assert pool.FUSED_SPLITK
    fused = _pool_run(pool, eng, hp, first, list(range(9)), K)
    pool2 = DecodePool(eng.llm, slots=64, slot_rows=512)
    pool2.FUSED_SPLITK = False
    plain = _pool_run(pool2, eng, hp, first, list(range(9)), K)
    same = sum(a == b for a, b in zip(fused, plain))
    assert same >= 7, f"only {same}/9 sequences decode to the same ids with and without the fused split-K consumers"


def _engine(seed=31):
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
    from vlm_fo1_amd.vit import ViTConfig
    cfg = FO1Config(vit=ViTConfig(depth=2, fullatt_block_indexes=(1,)), llm=LLMConfig(num_layers=2, vocab_size=4096, max_seq=1024))
    weights = random_weights(cfg, "cuda", seed=seed)
    weights["llm"]["embed_tokens.weight"] = (weights["llm"]["embed_tokens.weight"].float() * 4).bfloat16()   # real top-1 margins
    return cfg, weights, FO1Engine(cfg, weights, "cuda")


def _requests(n):
    from test_batched_prefill_gpu import make_request
    return [make_request(300 + i, 96 + 28 * (i % 4), 120 + 28 * (i % 3), 1 + (5 * i) % 9) for i in range(n)]


def _pool_run(pool, eng, hp, first, sel, K, stop=(), graph=True):
    """Join the selected sequences of the last prefill, decode to the end -> ids in `sel` order."""
    tags = [("t", b) for b in sel]
    pool.join(eng.llm.kcache, eng.llm.vtcache, [hp["seqs"][b] for b in sel], [hp["delta"][b] for b in sel],
              torch.stack([first[b] for b in sel]), K, stop, tags=tags)
    got = {tag[1]: ids for _, tag, ids in pool.drain(use_graph=graph, poll=3)}
    assert not pool.live and len(pool.free) == pool.P
    return [got[b] for b in sel]


@pytest.mark.parametrize("slots", [64, 128])
def test_pool_ids_independent_of_slot_neighbours_and_oracle(slots, product_library):
    from test_batched_decode_gpu import oracle_logits
    from vlm_fo1_amd.llm import DecodePool
    cfg, weights, eng = _engine()
    reqs = _requests(9)
    K = 10
    ref32 = eng.generate_batch(reqs, max_new_tokens=K, use_graph=True)          # the <= 32-sequence BatchDecoder
    eng.prefill_batch(reqs, use_graph=False)
    hp, first = eng._last_batch, eng._last_next_tokens.clone()
    pool = DecodePool(eng.llm, slots=slots)
    allg = _pool_run(pool, eng, hp, first, list(range(9)), K, graph=True)
    assert [len(t) for t in allg] == [K] * 9
    assert _pool_run(pool, eng, hp, first, list(range(9)), K, graph=False) == allg, "eager and graph-replayed pool steps differ"
    # alone (slot 0) and in another slot order: the same ids, bit for bit
    for b in (0, 4, 8):
        assert _pool_run(pool, eng, hp, first, [b], K) == [allg[b]], f"sequence {b} decodes differently alone than among 8 others"
    perm = [5, 2, 8, 0, 7, 1, 3, 6, 4]
    assert _pool_run(pool, eng, hp, first, perm, K) == [allg[b] for b in perm], "ids depend on the slot"
    tol = 0.05
    n_q = 0
    for b, r in enumerate(reqs[:4]):
        ref_ids, ref_logits = oracle_logits(cfg, weights, r, allg[b])
        for i, t in enumerate(allg[b]):
            top2 = ref_logits[i].topk(2).values
            assert float(ref_logits[i].max() - ref_logits[i][t]) <= 2 * tol, f"sequence {b} step {i}: pool token {t} is not (near-)optimal for the oracle"
            if float(top2[0] - top2[1]) > 2 * tol:
                n_q += 1
                assert t == ref_ids[i], f"sequence {b} step {i}: pool id {t} != oracle greedy id {ref_ids[i]}"
    assert n_q >= 4 * K // 2
    # the <= 32-sequence BatchDecoder: other kernels, other fp32 sum orders -> near-ties may differ, most sequences must not
    same = sum(int(a == b) for a, b in zip(ref32, allg))
    assert same >= 7, f"BatchDecoder: only {same}/9 sequences decode to the pool's ids"


def test_pool_stop_rule_budget_and_slot_reuse_mid_flight(product_library):
    from vlm_fo1_amd.llm import DecodePool
    cfg, weights, eng = _engine()
    reqs = _requests(6)
    K = 12
    eng.prefill_batch(reqs, use_graph=False)
    hp, first = eng._last_batch, eng._last_next_tokens.clone()
    pool = DecodePool(eng.llm, slots=64)
    full = _pool_run(pool, eng, hp, first, list(range(6)), K)
    stop = full[1][4]
    out = _pool_run(pool, eng, hp, first, list(range(6)), K, stop=[stop])
    for b, ids in enumerate(out):
        cut = full[b].index(stop) + 1 if stop in full[b] else K
        assert ids == full[b][:cut], f"sequence {b}: stop rule gave {ids}, expected {full[b][:cut]}"
    assert len(out[1]) <= 5
    assert _pool_run(pool, eng, hp, first, list(range(6)), 3) == [t[:3] for t in full], "budget"
    # slots re-used while others are mid-flight: 0..2 join, 4 steps, 3..5 join, 2 steps, ...; every sequence still gets its own ids
    pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][:3], hp["delta"][:3], first[:3], 6, (), tags=[0, 1, 2])
    for _ in range(4):
        pool.step()
    pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][3:], hp["delta"][3:], first[3:], K, (), tags=[3, 4, 5])
    got = {}
    for _ in range(3):
        pool.step()
    for _, tag, ids in pool.harvest(pool.snapshot()):
        got[tag] = ids
    assert sorted(got) == [0, 1, 2] and sorted(pool.free)[:3] == [0, 1, 2], "the 6-token sequences have finished and freed slots 0-2"
    pool.join(eng.llm.kcache, eng.llm.vtcache, hp["seqs"][:2], hp["delta"][:2], first[:2], K, (), tags=[10, 11])    # into the freed slots
    for _, tag, ids in pool.drain(poll=2):
        got[tag] = ids
    assert [got[b] for b in range(3)] == [t[:6] for t in full[:3]]
    assert [got[b] for b in (3, 4, 5)] == full[3:] and [got[10], got[11]] == full[:2]


def test_pool_mixes_submissions_with_different_stop_sets_and_budgets(product_library):
    """Round 5: stop-id sets are per SEQUENCE on the device (state[slot][6] -> a row of the pool's set table), so
    submissions with different stop rules and budgets decode TOGETHER — each sequence ends by its own rule, with the ids it gets alone."""
    from vlm_fo1_amd.llm import DecodePool
    cfg, weights, eng = _engine()
    reqs = _requests(6)
    K = 12
    eng.prefill_batch(reqs, use_graph=False)
    hp, first = eng._last_batch, eng._last_next_tokens.clone()
    pool = DecodePool(eng.llm, slots=64)
    full = _pool_run(pool, eng, hp, first, list(range(6)), K)
    stop_a, stop_b = full[0][3], full[4][5]
    kc, vt = eng.llm.kcache, eng.llm.vtcache
    pool.join(kc, vt, hp["seqs"][:2], hp["delta"][:2], first[:2], K, (stop_a,), tags=[0, 1])             # set A
    pool.join(kc, vt, hp["seqs"][2:4], hp["delta"][2:4], first[2:4], 7, (), tags=[2, 3])                  # no stop ids, budget 7
    pool.join(kc, vt, hp["seqs"][4:], hp["delta"][4:], first[4:], K, (stop_b, stop_a), tags=[4, 5])       # set B (contains A's id too)
    assert len(pool.live) == 6 and len({pool.slot_set[s] for s in pool.live}) == 3
    got = {tag: ids for _, tag, ids in pool.drain(poll=3)}

    def cut(ids, stops, budget):
        for i, t in enumerate(ids[:budget]):
            if t in stops:
                return ids[:i + 1]
        return ids[:budget]

    assert got[0] == cut(full[0], {stop_a}, K) and got[1] == cut(full[1], {stop_a}, K) and len(got[0]) == 4
    assert got[2] == full[2][:7] and got[3] == full[3][:7]
    assert got[4] == cut(full[4], {stop_a, stop_b}, K) and got[5] == cut(full[5], {stop_a, stop_b}, K) and len(got[4]) <= 6
    assert sum(pool._set_users) == 0 and not pool.live


def test_pool_service_from_replica_threads_equals_direct_pool_runs(product_library):
    """PoolService: three passes of three requests prefilled by two engine replicas on their own threads / streams, sequences of all
    passes decoding together; every request's ids == a direct DecodePool run of its pass (same prefill bits, slot-independent decode)."""
    from vlm_fo1_amd.llm import DecodePool
    cfg, weights, eng = _engine()
    reqs = _requests(9)
    passes = [reqs[0:3], reqs[3:6], reqs[6:9]]
    K = 9
    want = []
    direct = DecodePool(eng.llm, slots=64)
    for grp in passes:
        eng.prefill_batch(grp, use_graph=True)
        eng.prefill_batch(grp, use_graph=True)      # second sighting: the captured graph (what the service run replays)
        want.append(_pool_run(direct, eng, eng._last_batch, eng._last_next_tokens.clone(), [0, 1, 2], K))
    svc = eng.enable_decode_pool(slots=64, steps_per_round=2)
    try:
        engines = [eng, eng.replica()]
        assert engines[1]._pool_svc is svc
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got, errs = {}, []

        def worker(w):
            try:
                torch.cuda.set_device(0)
                with torch.cuda.stream(streams[w]):
                    hs = []
                    for rep in range(3):             # every pass three times: slots are freed and re-used while other passes decode
                        for p in range(w, 3, 2):
                            hs.append((rep, p, engines[w].submit_batch(passes[p], K, (), use_graph=True)))
                    for rep, p, h in hs:
                        got[(rep, p)] = h.result(timeout=300)
            except BaseException as e:
                errs.append(e)

        th = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for (rep, p), ids in got.items():
            assert ids == want[p], f"pass {p} (round {rep}) came back with other ids than the direct pool run"
        assert len(got) == 9 and svc.stats["finished"] == 27 and svc.stats["joined"] == 27
        # the blocking form: generate_batch routes through the pool too
        assert eng.generate_batch(passes[1], max_new_tokens=K, use_graph=True) == want[1]
    finally:
        eng.disable_decode_pool()


def test_pool_attention_one_chunk_writes_rows_without_combine(ab_library):
    """More than 32 sequences whose contexts fit ONE 1024-key chunk (the pool's default): the split kernel writes the normalised rows itself
    (no combine launch).  Against torch fp32 (P rounded to bf16 like the kernel), and against the 512-key split + combine form of the same
    kernel (other merge order: equal within a bf16 ulp); finished slots keep their previous rows."""
    from vlm_fo1_amd import lib as L, ops
    lib = L.load()
    B, H, KV, HD, slot = 40, 16, 2, 128, 1024
    scale = 1.0 / math.sqrt(HD)
    g = torch.Generator().manual_seed(5)
    kc = (torch.randn(KV, B * slot, HD, generator=g) * 0.5).bfloat16().cuda()
    vt = (torch.randn(KV * HD, B * slot, generator=g) * 0.5).bfloat16().cuda()
    q = (torch.randn(B, H * HD, generator=g) * 0.5).bfloat16().cuda()
    lens = [1 + (37 * b) % 700 for b in range(B)]
    lens[3], lens[7] = 1023, 512
    state = torch.zeros(B, 8, dtype=torch.int32)
    for b in range(B):
        state[b, 0] = b * slot + lens[b] - 1
        state[b, 2] = b * slot
    state[5, 3] = 1                                   # finished: skipped
    state = state.cuda()
    out = {}
    for chunk in (1024, 512):
        L.check(lib.fo1_attention_decode_set_pool_chunk(chunk), "chunk")
        o = ops.attention_decode_batch(q, kc, vt, state, 1024, H, KV, HD, scale)
        torch.cuda.synchronize()
        out[chunk] = o.float().cpu()
    L.check(lib.fo1_attention_decode_set_pool_chunk(1024), "chunk")
    L.check(lib.fo1_attention_decode_set_impl(2), "impl")          # A/B: four waves of a workgroup each walking their own tiles
    try:
        out["ws"] = ops.attention_decode_batch(q, kc, vt, state, 1024, H, KV, HD, scale).float().cpu()
        torch.cuda.synchronize()
    finally:
        lib.fo1_attention_decode_set_impl(0)
    grp = H // KV
    for b in range(B):
        if b == 5:
            continue
        k = kc[:, b * slot:b * slot + lens[b]].float().cpu()            # [KV, L, HD]
        v = vt[:, b * slot:b * slot + lens[b]].float().cpu().view(KV, HD, -1)
        qq = q[b].float().cpu().view(KV, grp, HD)
        s = torch.einsum("kgd,kld->kgl", qq, k) * scale
        p = torch.softmax(s, -1)
        ref = torch.einsum("kgl,kdl->kgd", p, v).reshape(H * HD)
        for c in (1024, 512, "ws"):
            err = (out[c][b] - ref).abs().max().item()
            assert err <= 2.0 ** -7 * ref.abs().max().item() + 2e-3, (b, c, err)
        assert (out[1024][b] - out[512][b]).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-3, b
