"""Multi-image varlen batched prefill (the reference's batch-aware splice, omchat_qwen2_5_vl.py:380-416): R images'
rows packed into ONE ViT / DaViT / SimpleFPN / LLM pass must give, per image, what the one-image pass gives.

  * With the GEMM tile pinned (the k-order of a row's dot products then does not depend on M) every output is BIT-identical to the
    sequential passes: packing changes which rows share a launch, not the arithmetic of a row.
  * With the automatic tile choice (large M takes the 256x256 32x32x16-MFMA kernel, whose fp32 summation order differs from the
    16x16x32 tiles) outputs agree at the bf16 tolerance of the other parity tests, and the greedy token is unchanged whenever
    the top-1 margin exceeds the logit tolerance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
KEYS = ("image_tokens", "region_tokens", "last_hidden", "logits", "next_token")


def make_engine(seed=7):
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
    from vlm_fo1_amd.vit import ViTConfig
    cfg = FO1Config(vit=ViTConfig(depth=2, fullatt_block_indexes=(1,)), llm=LLMConfig(num_layers=2, vocab_size=4096, max_seq=1024))
    return FO1Engine(cfg, random_weights(cfg, "cuda", seed=seed), "cuda")


def make_request(i, W, H, n):
    from vlm_fo1.model.image_processing import smart_resize
    from vlm_fo1_amd.model import synthetic_prompt
    rh, rw = smart_resize(H, W, 28, 56 * 56, 2048 * 2048)
    gh, gw = rh // 14, rw // 14
    g = torch.Generator().manual_seed(900 + i)
    b = torch.rand(n, 4, generator=g)
    x1, y1 = b[:, 0] * W * 0.7, b[:, 1] * H * 0.7
    boxes = torch.stack([x1, y1, x1 + 4 + b[:, 2] * W * 0.3, y1 + 4 + b[:, 3] * H * 0.3], 1)
    return dict(ids=synthetic_prompt(n, vocab=4096, seed=i), pix=torch.randn(gh * gw, 1176, generator=g).bfloat16().cuda(), grid=(gh, gw),
                aux=torch.randn(3, H, W, generator=g).bfloat16().cuda(), boxes=boxes.cuda())


def clone(o):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}


def test_ragged_batch_equals_sequential_bitwise_with_pinned_tile(ab_library):
    from vlm_fo1_amd import lib as L
    eng = make_engine()
    reqs = [make_request(0, 500, 399, 7), make_request(1, 333, 711, 33), make_request(2, 64, 60, 1), make_request(3, 420, 420, 100)]
    try:
        L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")     # 128x128 two-stage tiles for every GEMM, no split-K
        L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
        L.check(L.load().fo1_gemm_set_gemv(0), "gemv")              # M <= 4 (a tiny image's merger rows, the lm_head row) must not take the GEMV kernel in one run only
        seq = [clone(eng.prefill(r["ids"], r["pix"], r["grid"], r["aux"], r["boxes"])) for r in reqs]
        bat = eng.prefill_batch(reqs)
        for i, (a, b) in enumerate(zip(seq, bat)):
            for k in KEYS:
                assert torch.equal(a[k], b[k]), f"request {i}: {k} differs between the packed pass and the one-image pass"
            assert torch.equal(a["embeds"], b["embeds"]) and torch.equal(a["position_ids"], b["position_ids"]) and a["rope_delta"] == b["rope_delta"]
        # a different packing order permutes the results, nothing else
        perm = [2, 0, 3, 1]
        bat2 = eng.prefill_batch([reqs[j] for j in perm])
        for slot, j in enumerate(perm):
            for k in KEYS:
                assert torch.equal(bat2[slot][k], seq[j][k]), f"permuted batch: request {j} {k}"
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_gemv(1)


def test_uniform_batch_auto_tiles_tolerance_and_graph():
    """8 same-geometry images (the bench's shape of work, reduced depth): DaViT / FPN run stacked, the GEMMs take the large-M
    kernel.  Packed vs sequential at the bf16 tolerance; graph replay of the packed pass bit-identical to its eager launches."""
    eng = make_engine(seed=8)
    reqs = [make_request(10 + i, 640, 480, 12) for i in range(8)]
    seq = [clone(eng.prefill(r["ids"], r["pix"], r["grid"], r["aux"], r["boxes"])) for r in reqs]
    bat = [clone(o) for o in eng.prefill_batch(reqs)]
    for i, (a, b) in enumerate(zip(seq, bat)):
        # two bf16 executions of the same path with different fp32 summation orders: each sits at the bf16 noise floor of the stage
        # (tests/golden/bf16_floor.json: DaViT-L alone 0.9998), so their mutual deviation is bounded by about twice that
        for k, cmin in (("image_tokens", 0.9999), ("region_tokens", 0.9995), ("last_hidden", 0.9995)):
            x, y = a[k].float(), b[k].float()
            cos = F.cosine_similarity(x, y, dim=-1).min().item()
            rel = ((x - y).abs().max() / x.abs().max()).item()
            assert cos >= cmin and rel <= 2 ** -4, f"request {i} {k}: packed vs sequential min cos {cos:.6f} rel {rel:.4g}"
        err = (a["logits"].float() - b["logits"].float()).abs().max().item()
        assert err <= 0.05, f"request {i}: logits differ by {err:.4g}"
        top2 = a["logits"].float()[0].topk(2).values
        if top2[0] - top2[1] > 0.1:
            assert int(a["next_token"].item()) == int(b["next_token"].item())
    for _ in range(3):   # sighting 2 captures, 3 replays
        g = eng.prefill_batch(reqs, use_graph=True)
    for i, (b, c) in enumerate(zip(bat, g)):
        for k in KEYS:
            assert torch.equal(b[k], c[k]), f"request {i}: {k} differs between graph replay and eager (packed pass)"
    assert len(eng._graphs) == 1


def test_graph_cache_is_bounded_and_one_off_shapes_run_eagerly():
    """every new shape signature used to be captured and kept forever.  Now a signature is captured only once it
    repeats, and at most GRAPH_CACHE graphs are kept (LRU)."""
    eng = make_engine(seed=9)
    eng.GRAPH_CACHE = 2
    shapes = [(200, 160), (230, 120), (120, 260), (180, 180)]
    for i, (W, H) in enumerate(shapes):     # one-off shapes: nothing is captured
        r = make_request(40 + i, W, H, 3)
        eng.prefill(r["ids"], r["pix"], r["grid"], r["aux"], r["boxes"], use_graph=True)
    assert len(eng._graphs) == 0
    outs = {}
    for rep in range(2):                    # repeated shapes: captured on the second sighting, LRU of 2
        for i, (W, H) in enumerate(shapes):
            r = make_request(40 + i, W, H, 3)
            o = eng.prefill(r["ids"], r["pix"], r["grid"], r["aux"], r["boxes"], use_graph=True)
            if rep == 0:
                outs[i] = o["logits"].clone()
            else:
                assert torch.equal(o["logits"], outs[i])
    assert len(eng._graphs) == 2


def test_prompts_sharing_one_image_run_the_towers_once_and_match_separate_requests(ab_library):
    """BASELINE configs[4]: 300 proposals = 3 prompts of 100 over ONE image (the reference caps region features at 100 per prompt,
    mm_utils.py:600, and would run the whole model three times).  Requests with the same `image_id` share the image: ViT / DaViT /
    SimpleFPN run once, every prompt's <image> block reads the same token rows, HFRE pools each prompt's boxes on the shared maps.
    Per prompt the result is bit-identical to the same prompt sent as its own request (tile pinned), also mixed with another image."""
    from vlm_fo1_amd import lib as L
    eng = make_engine(seed=12)
    base = make_request(70, 420, 300, 30)
    other = make_request(71, 500, 399, 9)
    g = torch.Generator().manual_seed(5)
    from vlm_fo1_amd.model import synthetic_prompt
    prompts = []
    for k in range(3):
        b = base["boxes"][k * 10:(k + 1) * 10]
        prompts.append(dict(ids=synthetic_prompt(10, vocab=4096, seed=80 + k), pix=base["pix"], grid=base["grid"], aux=base["aux"], boxes=b))
    try:
        L.check(L.load().fo1_gemm_set_variant(2, 1), "variant")
        L.check(L.load().fo1_gemm_set_splitk(1), "splitk")
        L.check(L.load().fo1_gemm_set_gemv(0), "gemv")
        sep = [clone(eng.prefill_batch([r])[0]) for r in prompts + [other]]
        shared = [dict(r, image_id="img-a") for r in prompts]
        calls = []
        orig = eng.vit.forward_batch
        eng.vit.forward_batch = lambda pix, grids, capture="all": (calls.append(len(grids)), orig(pix, grids, capture))[1]
        got = eng.prefill_batch(shared + [other])
        eng.vit.forward_batch = orig
        assert calls == [2], f"two unique images in the pass, the ViT saw {calls}"
        for i, (a, b) in enumerate(zip(sep, got)):
            for k in KEYS:
                assert torch.equal(a[k], b[k]), f"prompt {i}: {k} differs between the shared-image pass and the separate request"
        # graph replay of the shared-image pass
        for _ in range(3):
            rep = eng.prefill_batch(shared + [other], use_graph=True)
        for a, b in zip(got, rep):
            for k in KEYS:
                assert torch.equal(a[k], b[k])
        with pytest.raises(ValueError):
            eng.prefill_batch([dict(prompts[0], image_id="x"), dict(other, image_id="x")])
    finally:
        L.load().fo1_gemm_set_variant(0, 0)
        L.load().fo1_gemm_set_splitk(0)
        L.load().fo1_gemm_set_gemv(1)


def test_shared_prefix_rows_run_through_the_llm_once(product_library):
    """Prompts over ONE image whose first rows are identical (system text + the image tokens, up to the first region) share those rows
    (llm.plan_batch(share_prefix=True), fo1_attention_prefix_bf16): the prefix runs through every layer once, each prompt's remaining
    rows attend [prefix | own rows].  In a causal model that is the same function as running every prompt's full rows (what the
    reference does, once per prompt: mm_utils.py:600 caps a prompt at 100 regions): compared here against the full-row pass of the
    same engine — equal up to the fp32 order of the online softmax (the key tiles of a prompt's own rows start at another offset) —
    through prefill, graph replay, the <= 32-sequence decoder and the decode pool."""
    import torch.nn.functional as F
    from vlm_fo1_amd.model import synthetic_prompt
    eng = make_engine(seed=13)
    img_a, img_b, other = make_request(72, 640, 480, 30), make_request(74, 500, 399, 20), make_request(73, 420, 300, 9)
    lead = synthetic_prompt(10, vocab=4096, seed=80)

    def prompt(k, n):      # the same leading text for every prompt of an image, another question at the end
        ids = synthetic_prompt(n, vocab=4096, seed=80)
        assert ids[:19] == lead[:19]
        return ids[:-3] + [1500 + k, 1600 + k, 1700 + k]

    reqs = [dict(ids=prompt(k, 10), pix=img_a["pix"], grid=img_a["grid"], aux=img_a["aux"], boxes=img_a["boxes"][k * 10:(k + 1) * 10], image_id="a") for k in range(3)]
    reqs += [other]
    reqs += [dict(ids=prompt(5 + k, 10), pix=img_b["pix"], grid=img_b["grid"], aux=img_b["aux"], boxes=img_b["boxes"][k * 10:(k + 1) * 10], image_id="b") for k in range(2)]
    eng.SHARE_PREFIX = False
    full = [clone(o) for o in eng.prefill_batch(reqs)]
    rows_full = eng._last_batch["rows"]
    ids_full = eng.generate_batch(reqs, max_new_tokens=6, use_graph=True)
    eng.SHARE_PREFIX = True
    got = [clone(o) for o in eng.prefill_batch(reqs)]
    hp = eng._last_batch
    assert [len(s) for s in hp["seqs"]] == [5, 5, 5, 3, 5, 5] and hp["seqs"][0][3] == hp["seqs"][2][3] and hp["seqs"][4][3] == hp["seqs"][5][3]
    P_a, P_b = hp["seqs"][0][4], hp["seqs"][4][4]
    assert P_a % 4 == 0 and P_a >= 391 and rows_full - hp["rows"] >= 2 * P_a + P_b - 16, (rows_full, hp["rows"], P_a, P_b)
    for i, (a, b) in enumerate(zip(full, got)):
        assert torch.equal(a["embeds"], b["embeds"]) and torch.equal(a["region_tokens"], b["region_tokens"]), f"prompt {i}: spliced rows differ"
        cos = F.cosine_similarity(a["last_hidden"].float(), b["last_hidden"].float(), dim=-1).item()
        d = (a["logits"].float() - b["logits"].float()).abs().max().item()
        assert cos >= 0.9995 and d <= 2.0 ** -6 * a["logits"].float().abs().max().item() + 1e-3, f"prompt {i}: cos {cos:.6f}, logits max |d| {d:.4g}"
    # graph replay == eager, bit for bit
    for _ in range(3):
        rep = eng.prefill_batch(reqs, use_graph=True)
    for a, b in zip(got, rep):
        for k in ("last_hidden", "logits", "next_token"):
            assert torch.equal(a[k], b[k])
    # decode from the shared layout (two-piece relocation): the <= 32-sequence decoder and the pool give the full-row pass's ids except at near-ties
    ids_shared = eng.generate_batch(reqs, max_new_tokens=6, use_graph=True)
    assert sum(int(a == b) for a, b in zip(ids_full, ids_shared)) >= 5, (ids_full, ids_shared)
    eng.enable_decode_pool(slots=64)
    try:
        ids_pool = eng.generate_batch(reqs, max_new_tokens=6, use_graph=True)
    finally:
        eng.disable_decode_pool()
    assert sum(int(a == b) for a, b in zip(ids_shared, ids_pool)) >= 5, (ids_shared, ids_pool)
