"""Seeded inputs of the splice parity cases (shared by tests/golden/make_splice_golden.py, which runs the reference's own
prepare_inputs_labels_for_qwen2_5_vl_multimodal on them, and tests/test_oracle_splice.py).  No reference import here."""
import torch

IMAGE, REGION = -200, -300            # vlm_fo1/constants.py IMAGE_TOKEN_INDEX / DEFAULT_REGION_INDEX
D, VOCAB = 32, 151700              # the table must hold <|vision_start|> / <|vision_end|> (151652 / 151653): get_rope_index finds images by them
IMAGE_TOKEN_ID, VISION_START, VISION_END, BOS = 151655, 151652, 151653, 151643


def embed_table():
    g = torch.Generator().manual_seed(11)
    return torch.randn(VOCAB, D, generator=g).bfloat16().float()


def prompt(n_sys, grid_merged, n_regions, n_tail, seed):
    """[sys text] <|vision_start|> <image> <|vision_end|> [text] (<tag id> <region>) x N [tail text] — the layout prepare_inputs
    produces (mm_utils.py:530-655); text ids < 151000."""
    g = torch.Generator().manual_seed(seed)
    r = lambda n: torch.randint(5, 151000, (n,), generator=g).tolist()
    ids = r(n_sys) + [VISION_START, IMAGE, VISION_END] + r(3)
    for _ in range(n_regions):
        ids += r(1) + [REGION]
    ids += r(n_tail)
    n_img = grid_merged[0] * grid_merged[1]
    img = torch.randn(n_img, D, generator=g).bfloat16().float()
    reg = torch.randn(max(n_regions, 1), D, generator=g).bfloat16().float()
    return dict(ids=ids, grid_merged=grid_merged, n_regions=n_regions, image_tokens=img, region_tokens=reg)


def cases():
    """name -> list of prompts (one batch each)."""
    return {
        "single_7_boxes": [prompt(14, (4, 6), 7, 9, 1)],
        "single_100_boxes": [prompt(18, (17, 23), 100, 30, 2)],
        "batch_of_3_ragged": [prompt(14, (4, 6), 5, 9, 3), prompt(20, (2, 2), 0, 3, 4), prompt(9, (6, 5), 12, 40, 5)],
    }
