"""Generates tests/golden/upn_ref.npz with the REFERENCE's own UPN modules (detect_tools/upn/models/*, ops/modules/ms_deform_attn.py)
imported in place from /root/reference (this container only) and run on the CPU in fp32 with the seeded weights / inputs of
tests/upn_cases.py.

What is stubbed to make the reference importable here (none of it carries arithmetic of the path):
  mmengine (Registry / build_from_cfg / Config: a 20-line registry), torchvision (import-time only), timm.models.layers
  (DropPath = identity at inference, to_2tuple, trunc_normal_), and the compiled `MultiScaleDeformableAttention` extension,
  whose forward is routed to the reference's own pure-torch ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:41-61).

usage: python tests/golden/make_upn_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import upn_cases as C  # noqa: E402

OUT = os.path.join(HERE, "upn_ref.npz")


def reference_upn():
    """-> the reference's `detect_tools.upn` package, importable on this CPU-only box."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Registry:
        def __init__(self, name):
            self.name, self.d = name, {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self.d[cls.__name__] = cls
                return cls
            return deco

        def get(self, k):
            return self.d[k]

    def build_from_cfg(cfg, registry, default_args=None):
        cfg = dict(cfg)
        return registry.get(cfg.pop("type"))(**cfg)

    class Config:
        @staticmethod
        def fromfile(path):
            ns = {}
            exec(open(path).read(), ns)
            return types.SimpleNamespace(**{k: v for k, v in ns.items() if not k.startswith("_")})

    stub("mmengine", Registry=Registry, build_from_cfg=build_from_cfg, Config=Config)
    tv = stub("torchvision")
    tv._is_tracing = lambda: False
    tv.ops = stub("torchvision.ops", nms=None)
    tv.transforms = stub("torchvision.transforms")
    tv.transforms.functional = stub("torchvision.transforms.functional")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    timm = stub("timm")
    timm.models = stub("timm.models")
    timm.models.layers = stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
                              trunc_normal_=torch.nn.init.trunc_normal_)
    msda = stub("MultiScaleDeformableAttention")
    for k in [k for k in sys.modules if k == "detect_tools" or k.startswith("detect_tools.")]:
        del sys.modules[k]                      # this repo's drop-in package must not shadow the reference's
    sys.path.insert(0, "/root/reference")
    import detect_tools.upn as U
    assert U.__file__.startswith("/root/reference"), U.__file__
    from detect_tools.upn.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    msda.ms_deform_attn_forward = lambda v, s, ls, loc, w, step: ms_deform_attn_core_pytorch(v, s, loc, w)
    return U


def main():
    U = reference_upn()
    out = {}
    # ---- deformable encoder, 2 layers, 5-level pyramid (models/encoder/upn_encoder.py) ----
    enc = U.build_encoder(dict(type="UPNEncoder", num_layers=2, d_model=C.D_MODEL, use_checkpoint=False, use_transformer_ckpt=False,
                               encoder_layer_cfg=dict(type="DeformableTransformerEncoderLayer", activation="relu", d_model=C.D_MODEL, dropout=0.0,
                                                      d_ffn=C.D_FFN, n_heads=C.N_HEADS, n_levels=C.N_LEVELS))).eval()
    missing, unexpected = enc.load_state_dict(C.encoder_state(2), strict=True), None
    src, pos = C.encoder_inputs()
    shapes = torch.as_tensor(C.ENC_SHAPES, dtype=torch.long)
    ls = torch.as_tensor(C.level_start(C.ENC_SHAPES), dtype=torch.long)
    with torch.no_grad():
        hooks, inter = [], {}
        hooks.append(enc.layers[0].self_attn.register_forward_hook(lambda m, i, o: inter.__setitem__("enc.layer0.self_attn", o.detach().clone())))
        hooks.append(enc.layers[0].register_forward_hook(lambda m, i, o: inter.__setitem__("enc.layer0", o.detach().clone())))
        mem = enc(src, pos, shapes, ls, torch.ones(1, C.N_LEVELS, 2), None)
        for h in hooks:
            h.remove()
    out["enc.memory"] = mem.numpy()
    out.update({k: v.numpy() for k, v in inter.items()})
    # ---- the whole deformable transformer + heads: 2 encoder + 2 decoder layers, 30 queries (deformable_transformer.py, upn_model.py) ----
    cfg = U.inference_wrapper.Config.fromfile("/root/reference/detect_tools/upn/configs/upn_large.py").model
    cfg["vision_backbone_cfg"]["backbone_cfg"] = "swin_T_224_1k"          # the backbone is bypassed below; a small one builds faster
    cfg["num_queries"] = cfg["transformer_cfg"]["num_queries"] = C.N_QUERIES_SMALL
    cfg["transformer_cfg"]["encoder_cfg"]["num_layers"] = 2
    cfg["transformer_cfg"]["decoder_cfg"]["num_layers"] = 2
    model = U.build_architecture(cfg).eval()
    st = C.transformer_state(2, 2, C.N_QUERIES_SMALL)
    res = model.load_state_dict(st, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    left = [k for k in res.missing_keys if not (k.startswith("backbone.") or k.startswith("input_proj.") or k == "transformer.level_embed"
                                                or "ref_point_head_point" in k or "bbox_embed." in k)]
    assert not left, left
    for i in range(1, 2):
        assert model.bbox_embed[i] is model.bbox_embed[0] and model.transformer.decoder.bbox_embed[i] is model.bbox_embed[0]
    src, pos = C.encoder_inputs(seed=78)
    rec = {}
    orig = model.transformer.get_two_stage_proposal

    def spy(memory, mask_flatten, spatial_shapes, ref_dict):
        rec["memory"] = memory.detach().clone()
        from detect_tools.upn.models.utils import gen_encoder_output_proposals
        om, op = gen_encoder_output_proposals(memory, mask_flatten, spatial_shapes, None)
        om = model.transformer.enc_output_norm(model.transformer.enc_output(om))
        rec["sel.scores"] = model.transformer.enc_out_class_embed(om, ref_dict).max(-1)[0].detach().clone()
        rec["sel.coords"] = (model.transformer.enc_out_bbox_embed(om) + op).detach().clone()
        r = orig(memory, mask_flatten, spatial_shapes, ref_dict)
        rec["sel.refpoints"] = r[0].detach().clone()
        return r

    model.transformer.get_two_stage_proposal = spy
    S = src.shape[1]
    model.forward_backbone_encoder = lambda samples: (src, pos, ls, shapes, torch.ones(1, C.N_LEVELS, 2), torch.zeros(1, S, dtype=torch.bool))
    dec_rec = {}
    hd = model.transformer.decoder.register_forward_hook(lambda m, i, o: dec_rec.__setitem__("o", o))
    with torch.no_grad():
        res = model(torch.zeros(1), "fine_grained_prompt")
    hd.remove()
    hs, refs = dec_rec["o"]
    out["tr.memory"] = rec["memory"].numpy()
    out["tr.sel.scores"] = rec["sel.scores"].numpy()
    coords = rec["sel.coords"].numpy().copy()
    out["tr.sel.coords"] = coords
    out["tr.sel.refpoints"] = rec["sel.refpoints"].numpy()
    out["tr.hs"] = torch.stack(hs).numpy()                      # [n_dec, 1, nq, 256] (decoder.norm applied)
    out["tr.refs"] = torch.stack(refs).numpy()                  # [n_dec + 1, 1, nq, 4] sigmoid space
    out["tr.pred_boxes"] = res["pred_boxes"].numpy()
    out["tr.pred_logits"] = res["pred_logits"].numpy()
    # ---- the whole detector: Swin-L widths at depths [2, 2, 2, 2] + input projections + 2/2-layer transformer, 100 x 136 image ----
    cfg = U.inference_wrapper.Config.fromfile("/root/reference/detect_tools/upn/configs/upn_large.py").model
    cfg["vision_backbone_cfg"]["backbone_cfg"] = dict(type="SwinTransformer", pretrain_img_size=384, embed_dim=C.SWIN_EMBED, depths=C.SWIN_DEPTHS_SMALL,
                                                      num_heads=C.SWIN_HEADS, window_size=C.SWIN_WINDOW, out_indices=(0, 1, 2, 3), dilation=False,
                                                      use_checkpoint=False)
    cfg["num_queries"] = cfg["transformer_cfg"]["num_queries"] = C.N_QUERIES_SMALL
    cfg["transformer_cfg"]["encoder_cfg"]["num_layers"] = 2
    cfg["transformer_cfg"]["decoder_cfg"]["num_layers"] = 2
    model = U.build_architecture(cfg).eval()
    res = model.load_state_dict(C.upn_state(), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    left = [k for k in res.missing_keys if not ("relative_position_index" in k or "ref_point_head_point" in k or "bbox_embed." in k)]
    assert not left, left
    from detect_tools.upn.models.module import nested_tensor_from_tensor_list
    img = C.test_image()
    feats = {}
    hk = model.backbone.model.backbone.register_forward_hook(lambda m, i, o: feats.update({k: v.tensors.detach().clone() for k, v in o.items()}))
    orig_fbe = model.forward_backbone_encoder
    fbe = {}

    def spy_fbe(samples):
        r = orig_fbe(samples)
        fbe["src"], fbe["pos"], fbe["shapes"] = r[0].detach().clone(), r[1].detach().clone(), r[3].clone()
        return r

    model.forward_backbone_encoder = spy_fbe
    with torch.no_grad():
        res = model(nested_tensor_from_tensor_list([img]), "fine_grained_prompt")
    hk.remove()
    for l in range(4):
        out[f"full.swin{l}"] = feats[l][0].flatten(1).t().contiguous().numpy().astype(np.float16)      # token-major [H*W, C]
    out["full.src"] = fbe["src"][0].numpy().astype(np.float16)
    out["full.pos"] = fbe["pos"][0].numpy().astype(np.float16)
    out["full.shapes"] = fbe["shapes"].numpy()
    out["full.pred_boxes"] = res["pred_boxes"][0].numpy()
    out["full.pred_logits"] = res["pred_logits"][0, :, 0].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
