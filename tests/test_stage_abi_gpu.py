"""Stage-level C-ABI entries (fo1_vit_forward / fo1_llm_prefill / fo1_llm_decode_step, csrc/stages.hip) against the Python
orchestration of the same primitives: bit-identical outputs — the entries issue the same launches in the same order — eager and
inside a captured hipGraph; and the whole engine run with every stage routed through them (FO1_STAGE_ABI) reproduces the default
engine's tokens, logits and generated ids exactly."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(depth=2, layers=2):
    from vlm_fo1_amd.llm import LLMConfig
    from vlm_fo1_amd.model import FO1Config, FO1Engine, random_weights
    from vlm_fo1_amd.vit import ViTConfig
    cfg = FO1Config(vit=ViTConfig(depth=depth, fullatt_block_indexes=(depth - 1,) if depth < 8 else (7, 15, 23, 31)),
                    llm=LLMConfig(num_layers=layers, vocab_size=4096, max_seq=2048))
    return cfg, FO1Engine(cfg, random_weights(cfg, "cuda", seed=11), "cuda")


def requests():
    from test_batched_prefill_gpu import make_request
    return [make_request(70, 500, 399, 7), make_request(71, 333, 711, 33), make_request(72, 96, 120, 2)]


@pytest.fixture()
def stage_switch():
    from vlm_fo1_amd import stage_abi
    old = stage_abi.ENABLED

    def set_(v):
        stage_abi.ENABLED = v
    yield set_
    stage_abi.ENABLED = old


def test_vit_forward_entry_bitwise(stage_switch):
    cfg, eng = build(depth=3)
    g = torch.Generator().manual_seed(3)
    grids = [(10, 14), (6, 8)]
    pix = torch.cat([torch.randn(a * b, 1176, generator=g) for a, b in grids]).bfloat16().cuda()
    stage_switch(False)
    t0, f0, _ = eng.vit.forward_batch(pix, grids, capture="all")
    stage_switch(True)
    t1, f1, _ = eng.vit.forward_batch(pix, grids, capture="all")
    assert torch.equal(t0, t1) and len(f0) == len(f1) and all(torch.equal(a, b) for a, b in zip(f0, f1))
    # captured: the entry is a pure launch sequence (fill kernels instead of memset nodes), replays stay identical
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        t2, f2, _ = eng.vit.forward_batch(pix, grids, capture="last")
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(t2, t0) and torch.equal(f2[-1], f0[-1])


def test_vit_forward_entry_takes_the_fused_qkv_epilogue_bitwise(stage_switch):
    """Four 640 x 480 images in one pass (6 256 patch rows: the q/k/v product runs on the 256 x 256 GEMM kernel): vit.py puts the 2-D RoPE and V^T
    into that GEMM's epilogue, and so does fo1_vit_forward given the head-major weight copy in its table (ABI 7) — bit-identical tokens and maps."""
    from vlm_fo1_amd import lib as L
    cfg, eng = build(depth=3)
    assert eng.vit._fused_qkv and eng.vit.blocks[0]["wqkv_hm"] is not None
    g = torch.Generator().manual_seed(6)
    grids = [(34, 46)] * 4
    S = sum(a * b for a, b in grids)
    assert L.load().fo1_gemm_takes_big_tile(S, 3 * cfg.vit.hidden_size, cfg.vit.hidden_size) == 1
    pix = torch.randn(S, 1176, generator=g).bfloat16().cuda()
    stage_switch(False)
    t0, f0, _ = eng.vit.forward_batch(pix, grids, capture="all")
    t0, f0 = t0.clone(), [f.clone() for f in f0]
    stage_switch(True)
    t1, f1, _ = eng.vit.forward_batch(pix, grids, capture="all")
    assert torch.equal(t0, t1) and len(f0) == len(f1) and all(torch.equal(a, b) for a, b in zip(f0, f1))
    # without the copy in the table the entry runs GEMM + fo1_qkv_post_vit_bf16: the same bits again
    from vlm_fo1_amd import stage_abi
    st = stage_abi.vit_stage(eng.vit) if hasattr(stage_abi, "vit_stage") else None
    if st is not None:
        saved = [(b.wqkv_hm, b.bqkv_hm) for b in st._blocks]
        for b in st._blocks:
            b.wqkv_hm = None
            b.bqkv_hm = None
        try:
            t2, f2, _ = eng.vit.forward_batch(pix, grids, capture="all")
            assert torch.equal(t0, t2) and all(torch.equal(a, b) for a, b in zip(f0, f2))
        finally:
            for b, (x, y) in zip(st._blocks, saved):
                b.wqkv_hm, b.bqkv_hm = x, y


def test_davit_fpn_projector_entries_bitwise(stage_switch):
    cfg, eng = build()
    g = torch.Generator().manual_seed(4)
    img = torch.randn(2, 3, 140, 196, generator=g).bfloat16().cuda()
    vmap = torch.randn(2 * 10 * 14, 1280, generator=g).bfloat16().cuda()
    feat = torch.randn(9, cfg.mm_region_hidden_size, generator=g).bfloat16().cuda()
    stage_switch(False)
    m0, s0 = eng.davit.forward(img)
    f0, fs0 = eng.fpn.forward(vmap, 10, 14, batch=2)
    p0 = eng.mm_projector_aux(feat)
    stage_switch(True)
    m1, s1 = eng.davit.forward(img)
    f1, fs1 = eng.fpn.forward(vmap, 10, 14, batch=2)
    p1 = eng.mm_projector_aux(feat)
    assert s0 == s1 and fs0 == fs1
    assert all(torch.equal(a, b) for a, b in zip(m0, m1)), "DaViT stage maps differ"
    assert all(torch.equal(a, b) for a, b in zip(f0, f1)), "SimpleFPN maps differ"
    assert torch.equal(p0, p1)
    m2, _ = eng.davit.forward(img[0].float())      # fp32 image, batch of one
    stage_switch(False)
    m3, _ = eng.davit.forward(img[0].float())
    assert all(torch.equal(a, b) for a, b in zip(m2, m3))


def test_davit_fpn_entries_take_the_implicit_convolution_bitwise(stage_switch):
    """Maps large enough for the 256 x 256 GEMM kernel: fo1_davit_forward / fo1_simplefpn_forward run the pre-norm ConvEmbed of stage 1 and the
    3x3 output convolutions of the two finest FPN levels as implicit GEMMs (index tables built on the device), like davit.py / fpn.py — and all
    three forms (stage entries, Python implicit, Python im2col) give the same bits."""
    import os
    from vlm_fo1_amd import lib as L
    cfg, eng = build()
    g = torch.Generator().manual_seed(5)
    img = torch.randn(5, 3, 480, 640, generator=g).bfloat16().cuda()
    vmap = torch.randn(2 * 34 * 46, 1280, generator=g).bfloat16().cuda()
    lib = L.load()
    assert lib.fo1_gemm_takes_big_tile(5 * 60 * 80, 512, 9 * 256) == 1            # DaViT stage 1 embed: 3x3 / stride 2 over 120 x 160 x 256
    assert lib.fo1_gemm_takes_big_tile(2 * 136 * 184, 512, 9 * 512) == 1          # FPN level 0 head at (4H, 4W)

    def run():
        m, s_ = eng.davit.forward(img)
        f, fs = eng.fpn.forward(vmap, 34, 46, batch=2)
        return [t.clone() for t in m], s_, [t.clone() for t in f], fs

    stage_switch(False)
    m0, s0, f0, fs0 = run()
    old = os.environ.get("FO1_CONV_IMPLICIT")
    os.environ["FO1_CONV_IMPLICIT"] = "0"
    try:
        m2, s2, f2, fs2 = run()
    finally:
        if old is None:
            del os.environ["FO1_CONV_IMPLICIT"]
        else:
            os.environ["FO1_CONV_IMPLICIT"] = old
    stage_switch(True)
    m1, s1, f1, fs1 = run()
    assert s0 == s1 == s2 and fs0 == fs1 == fs2
    for k, (a, b, c) in enumerate(zip(m0, m1, m2)):
        assert torch.equal(a, c), f"DaViT stage {k}: implicit vs im2col"
        assert torch.equal(a, b), f"DaViT stage {k}: python vs stage entry"
    for k, (a, b, c) in enumerate(zip(f0, f1, f2)):
        assert torch.equal(a, c), f"FPN level {k}: implicit vs im2col"
        assert torch.equal(a, b), f"FPN level {k}: python vs stage entry"


def test_engine_through_stage_entries_bitwise(stage_switch):
    cfg, eng = build()
    reqs = requests()
    stage_switch(False)
    ref = eng.prefill_batch(reqs, use_graph=False)
    ref_ids = eng.generate_batch(reqs, max_new_tokens=10, use_graph=False)
    keys = ("image_tokens", "region_tokens", "last_hidden", "logits")
    ref = [{k: o[k].clone() for k in keys} | {"next_token": int(o["next_token"])} for o in ref]
    stage_switch(True)
    got = eng.prefill_batch(reqs, use_graph=False)
    for a, b in zip(ref, got):
        for k in keys:
            assert torch.equal(a[k], b[k]), k
        assert a["next_token"] == int(b["next_token"])
    assert eng.generate_batch(reqs, max_new_tokens=10, use_graph=False) == ref_ids
    eng2 = eng.replica()    # fresh graph caches: captured with the stage entries inside
    assert eng2.generate_batch(reqs, max_new_tokens=10, use_graph=True) == ref_ids
    assert eng2.generate_batch(reqs, max_new_tokens=10, use_graph=True) == ref_ids


def test_llm_prefill_entry_takes_the_fused_qkv_epilogue_where_the_python_path_does(stage_switch):
    """A packed pass big enough for the 256 x 256 GEMM kernel (6 x ~650 rows >= 13 tile rows x 10 tile columns): both the Python orchestration
    and fo1_llm_prefill put mRoPE + K append + V^T in the q/k/v GEMM's epilogue (fo1_qkv_proj_rope_bf16) — bit-identical to each other, to the
    two-launch form (FO1_QKV_FUSED=0), and in the KV cache they leave behind (the decoded ids)."""
    import os
    from test_batched_prefill_gpu import make_request
    from vlm_fo1_amd import lib as L
    cfg, eng = build()
    reqs = [make_request(80 + i, 640, 480, 100) for i in range(6)]
    c = cfg.llm
    keys = ("last_hidden", "logits")

    def run():
        out = eng.prefill_batch(reqs, use_graph=False)
        ids = eng.generate_batch(reqs, max_new_tokens=6, use_graph=False)
        return [{k: o[k].clone() for k in keys} | {"next_token": int(o["next_token"])} for o in out], ids

    stage_switch(False)
    fused, ids_fused = run()
    rows = sum(len(r["ids"]) - 1 + r["grid"][0] * r["grid"][1] // 4 for r in reqs)      # a lower bound of the packed rows (one image sentinel each)
    assert rows >= 3328 and L.load().fo1_gemm_takes_big_tile(rows, (c.num_heads + 2 * c.num_kv_heads) * c.head_dim, c.hidden_size) == 1
    old = os.environ.get("FO1_QKV_FUSED")
    os.environ["FO1_QKV_FUSED"] = "0"
    try:
        plain, ids_plain = run()
    finally:
        if old is None:
            del os.environ["FO1_QKV_FUSED"]
        else:
            os.environ["FO1_QKV_FUSED"] = old
    stage_switch(True)
    staged, ids_staged = run()
    for a, b, d in zip(fused, plain, staged):
        for k in keys:
            assert torch.equal(a[k], b[k]), f"fused vs two-launch: {k}"
            assert torch.equal(a[k], d[k]), f"python vs stage entry: {k}"
        assert a["next_token"] == b["next_token"] == d["next_token"]
    assert ids_fused == ids_plain == ids_staged


def test_stage_entries_report_errors():
    import ctypes
    from vlm_fo1_amd import lib as L, stage_abi
    cfg, eng = build()
    st = stage_abi.llm_stage(eng.llm)
    need = L.load().fo1_llm_prefill_workspace_bytes(ctypes.byref(st.W), 64, 1)
    assert need > 64 * 1024 * 1024
    x = torch.zeros(64, cfg.llm.hidden_size, dtype=torch.bfloat16, device="cuda")
    C = st.cache_struct(eng.llm.kcache, eng.llm.vtcache)
    rc = L.load().fo1_llm_prefill(ctypes.byref(st.W), ctypes.byref(C), x.data_ptr(), x.stride(0), x.data_ptr(), x.data_ptr(), 64, 0, x.data_ptr(), 1, 64, 0.0,
                                  x.data_ptr(), 1, None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 1024, None)
    assert rc == -2 and b"workspace" in L.load().fo1_last_error()
