"""Host-side plan of the packed multi-prompt prefill (vlm_fo1_amd/llm.py plan_batch) — no GPU needed."""
import torch

from vlm_fo1_amd.llm import DEFAULT_REGION_INDEX, IMAGE_TOKEN_INDEX, LLMConfig, QwenLLM, rope_index_host


def _llm():
    m = object.__new__(QwenLLM)          # plan_batch / plan_inputs only read cfg
    m.cfg = LLMConfig(vocab_size=1000)
    return m


def test_plan_batch_layout():
    m = _llm()
    p0 = [5, 6, IMAGE_TOKEN_INDEX, 7, 11, DEFAULT_REGION_INDEX, 12, DEFAULT_REGION_INDEX, 8, 9]      # 2 regions
    p1 = [1, IMAGE_TOKEN_INDEX, 2, 3]                                                                  # no regions
    p2 = [4, 4, 4, IMAGE_TOKEN_INDEX, 21, DEFAULT_REGION_INDEX, 9]                                     # 1 region
    grids = [(2, 3), (1, 2), (2, 2)]
    n_img = [6, 2, 4]
    hp = m.plan_batch([p0, p1, p2], n_img, [2, 0, 1], grids)
    off = 0
    img0 = reg0 = 0
    for b, (ids, ni, nr, g) in enumerate(zip([p0, p1, p2], n_img, [2, 0, 1], grids)):
        o, L, Lp = hp["seqs"][b]
        assert o == off and o % 4 == 0 and Lp % 4 == 0 and 0 <= Lp - L < 4
        assert L == len(ids) - 1 + ni
        single, pos, delta = m.plan_inputs(ids, ni, nr, g)
        rows = hp["plan"][o:o + L]
        # same kinds; image / region indices shifted into the batch-concatenated tables
        assert torch.equal(rows[:, 0], single[:, 0])
        shift = (single[:, 0] == 1).int() * img0 + (single[:, 0] == 2).int() * reg0
        assert torch.equal(rows[:, 1], single[:, 1] + shift)
        assert torch.equal(hp["plan"][o + L:o + Lp], torch.zeros(Lp - L, 2, dtype=torch.int32))        # dummy rows: token 0
        assert torch.equal(hp["pos"][b], pos) and hp["delta"][b] == delta
        assert hp["last"][b].tolist() == [0, o + L - 1]
        ref_pos, ref_delta = rope_index_host(ids.index(IMAGE_TOKEN_INDEX), g, L - ids.index(IMAGE_TOKEN_INDEX) - ni)
        assert torch.equal(pos, ref_pos) and delta == ref_delta
        off += Lp
        img0 += ni
        reg0 += nr
    assert hp["rows"] == off and hp["cos"].shape == (off, 128) and hp["plan"].shape == (off, 2)


def test_plan_batch_share_prefix_reassembles_every_prompt():
    """llm.plan_batch(share_prefix=True): prompts over one image with a common leading part keep ONE copy of those rows (a multiple of
    4, at least SHARE_MIN_ROWS); prefix rows + own rows of every prompt are exactly the rows (gather plan, rope tables) of the
    unshared layout, the last-row plan points at the prompt's last real row, other prompts are untouched."""
    import torch
    from vlm_fo1_amd.llm import LLMConfig, QwenLLM
    from vlm_fo1_amd.model import synthetic_prompt
    llm = QwenLLM.__new__(QwenLLM)
    llm.cfg = LLMConfig()
    base = synthetic_prompt(100, n_text=60, seed=7)

    def variant(k, cut=0):
        ids = list(base)
        ids[-5] = 3000 + k
        return ids[:len(ids) - cut]

    prompts = [variant(0), variant(1, 3), variant(2), variant(0), variant(1), synthetic_prompt(5, seed=3), variant(9)]
    n_img, grids = [2304] * 5 + [391, 2304], [(48, 48)] * 5 + [(17, 23), (48, 48)]
    img_base = [0, 0, 0, 2304, 2304, 4608, 4608 + 391]
    n_reg = [100] * 5 + [5, 100]
    plain = llm.plan_batch(prompts, n_img, n_reg, grids, img_base=img_base)
    hp = llm.plan_batch(prompts, n_img, n_reg, grids, img_base=img_base, share_prefix=True)
    assert [len(s) for s in hp["seqs"]] == [5, 5, 5, 5, 5, 3, 3], "two groups share; the small image and the single prompt of the last image do not"
    assert hp["rows"] < plain["rows"] - 3 * 2300 and hp["seqs"][0][3] == hp["seqs"][1][3] == hp["seqs"][2][3] != hp["seqs"][3][3]
    for b, (s0, s1) in enumerate(zip(plain["seqs"], hp["seqs"])):
        o0, L, _ = s0
        if len(s1) == 5:
            o, L1, Lp, po, P = s1
            assert P % 4 == 0 and P >= llm.SHARE_MIN_ROWS and Lp % 4 == 0 and po + P <= o
            rows = torch.cat([hp["plan"][po:po + P], hp["plan"][o:o + L - P]])
            cs = torch.cat([hp["cos"][po:po + P], hp["cos"][o:o + L - P]])
            sn = torch.cat([hp["sin"][po:po + P], hp["sin"][o:o + L - P]])
            last = o + L - P - 1
        else:
            o, L1, Lp = s1
            rows, cs, sn, last = hp["plan"][o:o + L], hp["cos"][o:o + L], hp["sin"][o:o + L], o + L - 1
        assert L1 == L and torch.equal(rows, plain["plan"][o0:o0 + L]) and torch.equal(cs, plain["cos"][o0:o0 + L]) and torch.equal(sn, plain["sin"][o0:o0 + L])
        assert hp["last"][b].tolist() == [0, last] and hp["delta"][b] == plain["delta"][b] and torch.equal(hp["pos"][b], plain["pos"][b])
    from vlm_fo1_amd.llm import reloc_rows
    rr = reloc_rows(hp["seqs"][:2] + hp["seqs"][5:6], [0, 4096, 8192])
    (o, L, _, po, P), (o1, L1, *_), (o5, L5, _) = hp["seqs"][0], hp["seqs"][1], hp["seqs"][5]
    assert rr == [[po, 0, P, 0], [o, P, L - P, 0], [po, 4096, P, 0], [o1, 4096 + P, L1 - P, 0], [o5, 8192, L5, 0]]


def test_conv_plan_outlives_its_cache_entry_through_the_pass_keep_list():
    """A packed pass (and a hipGraph captured from it) holds a convolution plan's device tables by raw pointer: the plan must stay
    alive through the pass's keep list after the bounded plan cache has evicted it (ops.keep_scope / ops.keep_alive)."""
    from vlm_fo1_amd import ops
    keep = []
    with ops.keep_scope(keep):
        pl = ops.conv3x3_plan(((6, 5),) * 2, 1, 64, "cpu")
        assert ops.conv3x3_plan(((6, 5),) * 2, 1, 64, "cpu") is pl
    assert keep == [pl]                                            # once, however often the pass asked for it
    assert pl.M_in == 60 and pl.M_out == 60 and pl.Wp == 7 and pl.pad_rows == 2 * 8 * 7
    # output pixel (0, 0) of image 1 reads its taps from padded row 8 * 7 on: byte offset = that row * Cin * 2
    assert int(pl.a_rows[30].item()) == 8 * 7 * 64 * 2 and int(pl.rowmap[0].item()) == 7 + 1
    for i in range(70):                                            # push the entry out of the 64-entry cache
        ops.conv3x3_plan(((3 + i, 4),), 1, 64, "cpu")
    assert ops.conv3x3_plan(((6, 5),) * 2, 1, 64, "cpu") is not pl  # evicted and rebuilt ...
    assert keep[0] is pl and pl.rowmap.numel() == 60               # ... while the pass's list still owns the old tables
    ops.keep_alive(object())                                       # outside a scope: a no-op
    assert len(keep) == 1


def test_split_passes_respects_request_and_row_budgets_and_keeps_order():
    """FO1Engine.split_passes (host logic): <= PREFILL_MAX requests and <= PREFILL_ROWS ViT patch rows per packed pass, prompts over ONE image
    count its rows once, a request larger than the budget is its own pass, request order is kept."""
    from types import SimpleNamespace
    from vlm_fo1_amd.model import FO1Engine
    eng = SimpleNamespace(PREFILL_MAX=4, PREFILL_ROWS=1000)
    split = lambda reqs: FO1Engine.split_passes(eng, reqs)
    r = lambda k, gh, gw, image_id=None: dict(k=k, grid=(gh, gw), image_id=image_id)
    assert split([]) == []
    # request budget
    out = split([r(i, 10, 10) for i in range(9)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1, 2, 3], [4, 5, 6, 7], [8]]
    # row budget: 400 + 400 fit, the third 400 opens a new pass; a 2 000-row request is alone
    out = split([r(0, 20, 20), r(1, 20, 20), r(2, 20, 20), r(3, 40, 50), r(4, 10, 10)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1], [2], [3], [4]]
    # three prompts over one image: its 900 rows count once, so a 100-row image still fits beside them
    out = split([r(0, 30, 30, "a"), r(1, 30, 30, "a"), r(2, 30, 30, "a"), r(3, 10, 10)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1, 2, 3]]
    # ... and when the group is cut by the request budget the image's rows are counted again in the next pass
    eng.PREFILL_MAX = 2
    out = split([r(0, 30, 30, "a"), r(1, 30, 30, "a"), r(2, 30, 30, "a"), r(3, 20, 20)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1], [2], [3]]


def test_split_passes_budgets_aux_pixels_too():
    """the DaViT / SimpleFPN scratch follows the AUX image size (native resolution in `dynamic` mode), so small-grid images with large
    aux tensors must not share one oversized pass; prompts over one image count its aux pixels once."""
    from types import SimpleNamespace
    import torch
    from vlm_fo1_amd.model import FO1Engine
    eng = SimpleNamespace(PREFILL_MAX=8, PREFILL_ROWS=10 ** 6, PREFILL_AUX_PIXELS=1000 * 1000)
    split = lambda reqs: FO1Engine.split_passes(eng, reqs)
    r = lambda k, h, w, image_id=None: dict(k=k, grid=(4, 4), aux=torch.empty(3, h, w, device="meta"), image_id=image_id)
    out = split([r(0, 600, 800), r(1, 600, 800), r(2, 600, 800), r(3, 2000, 2000), r(4, 100, 100)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1], [2], [3], [4]]
    out = split([r(0, 900, 1000, "a"), r(1, 900, 1000, "a"), r(2, 900, 1000, "a"), r(3, 300, 300)])
    assert [[q["k"] for q in g] for g in out] == [[0, 1, 2, 3]]
    # an engine without the budget (older callers' stand-ins) keeps the two other budgets only
    del eng.PREFILL_AUX_PIXELS
    assert len(split([r(i, 2000, 2000) for i in range(4)])) == 1


def test_attention_32x32_form_is_not_chosen_past_its_32_bit_offsets():
    """q_block 128 / 256 address key rows with 32-bit byte offsets from the K base; segments ending past 4 GiB / 8 KB rows take the
    16x16 kernel, and a hand-built item list on such an operand fails loudly."""
    import pytest
    from vlm_fo1_amd import ops
    assert ops.pick_q_block([(0, 1564)], 16, 80) == 256
    far = 2 ** 32 // ops.ATTN32_MAX_ROW_BYTES
    assert ops.pick_q_block([(far - 2000, far - 1)], 16, 80) == 256
    assert ops.pick_q_block([(far - 2000, far)], 16, 80) == 64
    assert ops.pick_q_block([(far, far + 700)], 16, 128, 2) == 64
    ops._check_attn32_extent(64, far * 4, 8192)
    ops._check_attn32_extent(256, far - 1, 8192)
    with pytest.raises(ValueError, match="32-bit"):
        ops._check_attn32_extent(256, far, 8192)


def test_attention_work_list_block_choice_and_lpt_order(monkeypatch):
    """ops.pick_q_block / order_items / make_items (host logic of the attention work lists, round 5)."""
    import torch
    from vlm_fo1_amd import ops
    monkeypatch.delenv("FO1_ATTN32", raising=False)
    # LLM prefill: head dim 128, grouped-query (16 q heads on 2 kv heads) -> 128 queries x the 2 heads of a kv head
    assert ops.pick_q_block([(0, 651)], 16, 128, 2) == 128
    assert ops.pick_q_block([(0, 64)], 16, 128, 2) == 64             # a segment of one key tile stays on the 16x16 kernel
    # ViT: head dim 80, no grouping -> 256 queries of one head for full attention, 64 for the 64-token windows
    assert ops.pick_q_block([(0, 1564)], 16, 80) == 256
    assert ops.pick_q_block([(0, 64), (64, 128)], 16, 80) == 64
    assert ops.pick_q_block([(0, 5000)], 8, 32) == 64                # DaViT's head dim: always the 16x16 kernel
    monkeypatch.setenv("FO1_ATTN32", "0")
    assert ops.pick_q_block([(0, 651)], 16, 128, 2) == 64
    monkeypatch.delenv("FO1_ATTN32")
    # items: every query exactly once, blocks within their segment, longest walk first when causal (LPT over the launch)
    segs = [(0, 652), (652, 1304), (1304, 1400)]
    it = ops.make_items(segs, "cpu", causal=True, block=128)
    assert it.q_block == 128 and it.dtype == torch.int32
    rows = it.tolist()
    covered = sorted(q for q0, q1, _, _ in rows for q in range(q0, q1))
    assert covered == list(range(1400))
    assert all(k0 <= q0 < q1 <= k1 and q1 - q0 <= 128 and (k0, k1) in segs for q0, q1, k0, k1 in rows)
    tiles = [(q1 - k0 + 63) // 64 for q0, q1, k0, k1 in rows]
    assert tiles == sorted(tiles, reverse=True)
    # the 16x16 kernel's lists keep segment order (its grid is not walked longest-first)
    it64 = ops.make_items(segs, "cpu", causal=True, block=64)
    assert it64.tolist()[0][:2] == [0, 64] and it64.tolist()[-1][1] == 1400
    # a second key range (shared prefix) counts in the walk
    order = ops.order_items([[0, 128, 0, 128], [128, 256, 128, 256]], 128, True, prefix=[[0, 0], [0, 4096]])
    assert order == [1, 0]


def test_conv_plan_tables_address_exactly_the_im2col_matrix():
    """ops.Conv3x3Plan on the host: writing a map's pixels to rowmap[...] of the zero-framed buffer and reading the nine taps of output pixel m
    at a_rows[m] + (ky * Wp + kx) * Cin * 2 bytes reproduces F.unfold(pad 1, kernel 3) — for stride 1 and 2, and for images of different sizes
    packed into one buffer with a common row pitch."""
    import torch
    import torch.nn.functional as F
    from vlm_fo1_amd import ops
    g = torch.Generator().manual_seed(12)
    cin = 64
    for sizes, stride in ((((5, 7),), 1), (((6, 8), (6, 8)), 2), (((4, 9), (7, 3), (5, 5)), 1), (((9, 4), (3, 11)), 2)):
        pl = ops.conv3x3_plan(sizes, stride, cin, "cpu")
        maps = [torch.randn(h, w, cin, generator=g) for h, w in sizes]
        xpad = torch.zeros(pl.pad_rows, cin)
        xpad[pl.rowmap.long()] = torch.cat([m.reshape(-1, cin) for m in maps])                       # what layernorm_rows does with its output rows
        a = (pl.a_rows.numpy().view("uint32").astype("int64") // (cin * 2))                          # tap-0 row of every output pixel
        assert (pl.a_rows.numpy().view("uint32").astype("int64") % (cin * 2) == 0).all()
        col = torch.stack([xpad[torch.from_numpy(a + ky * pl.Wp + kx)] for ky in range(3) for kx in range(3)], 1)   # [M_out, 9, cin]: K order (ky, kx, c)
        ref = []
        for m in maps:
            u = F.unfold(m.permute(2, 0, 1)[None], kernel_size=3, padding=1, stride=stride)[0]       # [cin * 9, Ho * Wo], rows (c, ky, kx)
            ref.append(u.reshape(cin, 9, -1).permute(2, 1, 0))                                       # -> [Ho * Wo, 9, cin]
        ref = torch.cat(ref)
        assert col.shape == ref.shape == (pl.M_out, 9, cin)
        assert torch.equal(col, ref), (sizes, stride)
        assert pl.out_hw == [((h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1) for h, w in sizes]
