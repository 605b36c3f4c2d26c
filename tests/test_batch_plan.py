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
