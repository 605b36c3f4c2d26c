"""Generates tests/golden/{vit,davit,fpn,llm}_ref.npz: outputs of the REFERENCE's own modules (imported in place from
/root/reference — the LLM too: the vendored Qwen2_5_VLModel, see llm()) at TRUE channel widths and reduced depth, fp32 on the CPU, on seeded inputs.  Weights are NOT stored: they come from
the CPU-seeded `random_*_state` helpers of oracle/, which reproduce bit-identically anywhere, so the goldens hold outputs only.
tests/test_golden_towers.py (CPU: oracle vs golden) and tests/test_golden_towers_gpu.py (HIP engine vs golden) consume them —
neither needs /root/reference.

    python tests/golden/make_tower_goldens.py
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_tower_cases import DAVIT, FPN, LLM, VIT, davit_input, fpn_input, llm_input, vit_input  # noqa: E402
from oracle import davit_oracle as DO, fpn_oracle as FO, hfre_oracle as HO, llm_oracle as LO, reference_loader as R, vit_oracle as VO  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
noop = lambda t, *a, **k: t


def vit():
    qwen, enc = R.vendored_qwen(), R.vendored_vit_encoder()
    c = VIT
    cfg = qwen.Qwen2_5_VLVisionConfig(depth=c["depth"], hidden_size=1280, hidden_act="silu", intermediate_size=3420, num_heads=16,
                                      in_channels=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
                                      out_hidden_size=2048, fullatt_block_indexes=list(c["fullatt"]))
    cfg._attn_implementation = "sdpa"
    model = qwen.Qwen2_5_VisionTransformerPretrainedModel._from_config(cfg, attn_implementation="sdpa").eval().float()
    sd = VO.random_vit_state(c["depth"], 1280, 16, 3420, 2048, seed=c["seed"])
    model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    gh, gw = c["grid"]
    pix = vit_input()
    gather = enc.VisionFeaturesGather()
    model.vision_features_gather = gather
    with torch.no_grad():
        tokens = enc.custom_forward(model, pix.float(), torch.tensor([[1, gh, gw]]))
        maps = gather.extract_multi_level_features()[0]
    np.savez_compressed(os.path.join(OUT, "vit_ref.npz"), tokens=tokens.numpy(),
                        last_map=maps[-1][0].permute(1, 2, 0).reshape(gh * gw, 1280).numpy())
    print("vit", tuple(tokens.shape))


def davit():
    dv, cfgs = R.vendored_davit()
    cfg = cfgs.model_configs["davit-large"]
    with mock.patch.object(dv, "trunc_normal_", noop), mock.patch("torch.nn.init.normal_", noop), \
            mock.patch("torch.nn.init.kaiming_uniform_", noop), mock.patch("torch.nn.init.uniform_", noop), \
            mock.patch("torch.nn.init.constant_", noop):
        m = dv.DaViT(depths=cfg["depths"], embed_dims=cfg["dim_embed"], num_heads=cfg["num_heads"], num_groups=cfg["num_groups"],
                     patch_size=cfg["patch_size"], patch_stride=cfg["patch_stride"], patch_padding=cfg["patch_padding"],
                     patch_prenorm=cfg["patch_prenorm"], window_size=cfg["window_size"]).eval()
    sd = DO.random_davit_state(DO.DAVIT_LARGE, seed=DAVIT["seed"])
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    img = davit_input()
    with torch.no_grad():
        ref = m.forward_features(img.float())["image_features"]
    np.savez_compressed(os.path.join(OUT, "davit_ref.npz"),
                        **{f"stage{i}": r[0].permute(1, 2, 0).reshape(-1, r.shape[1]).numpy() for i, r in enumerate(ref)},
                        sizes=np.array([r.shape[2:] for r in ref]))
    print("davit", [tuple(r.shape) for r in ref])


def fpn():
    _, SimpleFP, _ = HO.load_reference_hfre()
    m = SimpleFP(out_channels=512, norm="LN", square_pad=0, dim=1280, stride=14).eval()
    sd = FO.random_fpn_state(seed=FPN["seed"])
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    gh, gw = FPN["grid"]
    x = fpn_input()
    with torch.no_grad():
        ref = m(x.float().reshape(gh, gw, 1280).permute(2, 0, 1).unsqueeze(0))
    np.savez_compressed(os.path.join(OUT, "fpn_ref.npz"), **{f"level{i}": r[0].permute(1, 2, 0).reshape(-1, 512).numpy() for i, r in enumerate(ref)})
    print("fpn", [tuple(r.shape) for r in ref])


def llm():
    """The reference's OWN vendored Qwen2_5_VLModel (modeling_qwen2_5_vl.py:1097-1242) run in place: oracle/reference_loader.py:vendored_llm
    builds it under the installed transformers with two construction shims (rope initialiser key, pad_token_id), none in the arithmetic."""
    c = LLM
    m = R.vendored_llm(vocab_size=c["vocab"], hidden_size=2048, intermediate_size=11008, num_hidden_layers=c["layers"],
                       num_attention_heads=16, num_key_value_heads=2, max_position_embeddings=4096, rms_norm_eps=1e-6,
                       rope_theta=1e6, rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, tie_word_embeddings=True)
    sd = LO.random_llm_state(c["layers"], 2048, 16, 2, 128, 11008, c["vocab"], seed=c["seed"])
    res = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    x, pos = llm_input()
    with torch.no_grad():
        ref = m(inputs_embeds=x.float()[None], position_ids=pos[:, None, :]).last_hidden_state[0]
    logits = ref[-1:] @ sd["embed_tokens.weight"].float().t()
    np.savez_compressed(os.path.join(OUT, "llm_ref.npz"), hidden=ref.numpy(), last_logits=logits.numpy())
    print("llm", tuple(ref.shape))


if __name__ == "__main__":
    assert R.available(), "/root/reference is needed to generate the goldens"
    torch.set_num_threads(8)
    only = sys.argv[1:] or ["vit", "davit", "fpn", "llm"]
    for name in only:
        {"vit": vit, "davit": davit, "fpn": fpn, "llm": llm}[name]()
