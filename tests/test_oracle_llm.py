"""CPU tests pinning oracle/llm_oracle.py: against the reference's own vendored Qwen2_5_VLModel run in place (prefill and KV-cache
decode; oracle/reference_loader.py:vendored_llm), against the installed HF Qwen2_5_VLTextModel, and get_rope_index against the
reference's own vendored function called unbound (it is pure index arithmetic)."""
import types

import pytest
import torch

from oracle import llm_oracle as LO
from oracle import reference_loader as R


def tiny_cfg():
    return dict(n_layers=2, n_heads=2, n_kv=1, head_dim=128, eps=1e-6, theta=1e6, sections=(16, 24, 24))


def test_llm_oracle_matches_hf_text_model():
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig
    cfg = Qwen2_5_VLTextConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                               num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=1024,
                               rms_norm_eps=1e-6, rope_theta=1e6, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                               rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, tie_word_embeddings=True)
    m = M.Qwen2_5_VLTextModel(cfg).eval()
    sd = LO.random_llm_state(2, 256, 2, 1, 128, 512, 512, seed=3)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    L = 40
    torch.manual_seed(0)
    x = torch.randn(L, 256).bfloat16().float()
    pos, _ = LO.rope_index(7, (3, 5), L - 7 - 15)
    with torch.no_grad():
        ref = m(inputs_embeds=x[None], position_ids=pos[:, None, :]).last_hidden_state[0]
    got = LO.llm_forward(sd, x, pos, bf16_rope_tables=False, **tiny_cfg())
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)


def _vendored_tiny():
    return R.vendored_llm(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                          num_key_value_heads=1, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=1e6,
                          rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, tie_word_embeddings=True)


@pytest.mark.skipif(not R.available(), reason="/root/reference not present")
def test_llm_oracle_matches_the_reference_s_own_vendored_model():
    """The oracle against the reference's vendored Qwen2_5_VLModel (modeling_qwen2_5_vl.py:1097-1242) run in place on the CPU — two
    construction shims for transformers 5, none in the arithmetic (oracle/reference_loader.py:vendored_llm)."""
    m = _vendored_tiny()
    sd = LO.random_llm_state(2, 256, 2, 1, 128, 512, 512, seed=3)
    res = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for L, nb, grid in ((40, 7, (3, 5)), (95, 20, (5, 7))):
        torch.manual_seed(L)
        x = torch.randn(L, 256).bfloat16().float()
        pos, _ = LO.rope_index(nb, grid, L - nb - grid[0] * grid[1])
        with torch.no_grad():
            ref = m(inputs_embeds=x[None], position_ids=pos[:, None, :]).last_hidden_state[0]
        got = LO.llm_forward(sd, x, pos, bf16_rope_tables=False, **tiny_cfg())
        torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.skipif(not R.available(), reason="/root/reference not present")
def test_llm_oracle_kv_cache_decode_matches_the_vendored_model():
    """Greedy continuation through the KV cache: the vendored model's own 1-token path (cache_position + rope delta,
    omchat_qwen2_5_vl.py:143-155) against the oracle's cached decode, hidden state per step."""
    m = _vendored_tiny()
    sd = LO.random_llm_state(2, 256, 2, 1, 128, 512, 512, seed=4)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    L, nb, grid = 40, 7, (3, 5)
    torch.manual_seed(1)
    x = torch.randn(L, 256).bfloat16().float()
    pos, delta = LO.rope_index(nb, grid, L - nb - grid[0] * grid[1])
    emb = sd["embed_tokens.weight"].float()
    with torch.no_grad():
        out = m(inputs_embeds=x[None], position_ids=pos[:, None, :], use_cache=True)
        past = out.past_key_values
        hid, cache = LO.llm_forward_cached(sd, x, pos, None, bf16_rope_tables=False, **tiny_cfg())   # (the fp32 model keeps fp32 tables)
        torch.testing.assert_close(hid, out.last_hidden_state[0], rtol=2e-5, atol=2e-5)
        tok = 11
        for step in range(4):
            p = L + step + delta
            pid = torch.full((3, 1, 1), p, dtype=torch.long)
            out = m(inputs_embeds=emb[tok:tok + 1][None], position_ids=pid, past_key_values=past, use_cache=True)
            past = out.past_key_values
            hid, cache = LO.llm_forward_cached(sd, emb[tok:tok + 1], torch.full((3, 1), p, dtype=torch.long), cache, bf16_rope_tables=False, **tiny_cfg())
            torch.testing.assert_close(hid, out.last_hidden_state[0], rtol=2e-5, atol=2e-5)
            tok = int((hid[-1] @ emb.t()).argmax())


@pytest.mark.skipif(not R.available(), reason="/root/reference not present")
@pytest.mark.parametrize("n_before,grid,n_after", [(18, (17, 23), 250), (3, (2, 2), 1), (30, (48, 48), 700)])
def test_rope_index_matches_reference(n_before, grid, n_after):
    ref = R.vendored_qwen()
    gh, gw = grid
    IMG, VSTART = 151655, 151652
    ids = [5] * (n_before - 1) + [VSTART] + [IMG] * (gh * gw) + [7] * n_after
    fake = types.SimpleNamespace(config=types.SimpleNamespace(
        vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=2),
        image_token_id=IMG, video_token_id=151656, vision_start_token_id=VSTART))
    pos_ref, delta_ref = ref.Qwen2_5_VLForConditionalGeneration.get_rope_index(
        fake, torch.tensor([ids]), torch.tensor([[1, gh * 2, gw * 2]]), None, None, torch.ones(1, len(ids), dtype=torch.long))
    pos, delta = LO.rope_index(n_before, grid, n_after)
    assert torch.equal(pos, pos_ref[:, 0])
    assert delta == int(delta_ref.item())
    # the engine's host-side copy of the same arithmetic
    from vlm_fo1_amd.llm import rope_index_host
    p2, d2 = rope_index_host(n_before, grid, n_after)
    assert torch.equal(p2, pos) and d2 == delta


def test_mrope_tables_match_engine_host_code():
    from vlm_fo1_amd.llm import mrope_tables
    pos, _ = LO.rope_index(11, (4, 6), 9)
    c, s = LO.mrope_cos_sin(pos, 128, 1e6, (16, 24, 24))
    c2, s2 = mrope_tables(pos, 128, 1e6, (16, 24, 24))
    assert torch.equal(c.bfloat16(), c2) and torch.equal(s.bfloat16(), s2)


def test_splice_layout():
    emb = torch.arange(10 * 4, dtype=torch.float32).reshape(10, 4)
    img = torch.full((6, 4), -1.0)
    reg = torch.stack([torch.full((4,), 100.0 + i) for i in range(2)])
    ids = torch.tensor([1, 2, LO.IMAGE_TOKEN_INDEX, 3, LO.DEFAULT_REGION_INDEX, 4, LO.DEFAULT_REGION_INDEX, 5])
    out, nb, na = LO.splice(ids, emb, img, reg)
    assert out.shape[0] == 7 + 6 and nb == 2 and na == 5
    assert torch.equal(out[2:8], img) and out[9, 0] == 100 and out[11, 0] == 101 and torch.equal(out[12], emb[5])
