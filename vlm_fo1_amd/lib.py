"""ctypes binding of libfo1hip.so (include/fo1.h).

The library is the product: if it is missing or fails to load this module raises —
there is no eager/CPU fallback anywhere in the package."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfo1hip.so")          # product library (include/fo1.h)
LIB_PATH_AB = os.path.join(_HERE, "libfo1hip_ab.so")    # test / bench build (+ include/fo1_ab.h), loaded when FO1_AB=1

c_int, c_float, c_void_p, c_size_t, c_int32, c_longlong = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                                           ctypes.c_size_t, ctypes.c_int32, ctypes.c_longlong)


class HfreSource(ctypes.Structure):
    """fo1_hfre_source_t (include/fo1.h)."""
    _fields_ = [
        ("data", c_void_p),
        ("H", c_int32), ("W", c_int32), ("C", c_int32), ("ld", c_int32),
        ("roi_H", c_int32), ("roi_W", c_int32),
        ("spatial_scale", c_float),
        ("box_space", c_int32),
        ("out_offset", c_int32),
    ]


class HfreOpts(ctypes.Structure):
    """fo1_hfre_opts_t (include/fo1.h)."""
    _fields_ = [
        ("batch", c_int32), ("box_image", c_void_p), ("img_stride", c_longlong * 8),
        ("ln_on", c_int32), ("ln_split", c_int32),
        ("ln_w0", c_void_p), ("ln_b0", c_void_p), ("ln_w1", c_void_p), ("ln_b1", c_void_p),
        ("ln_eps", c_float),
        ("out_bf16", c_void_p), ("out_bf16_ld", c_int32),
    ]


class VitBlock(ctypes.Structure):
    """fo1_vit_block_t"""
    _fields_ = [(n, c_void_p) for n in ("n1", "n2", "wqkv", "bqkv", "wo", "bo", "wgu", "bgu", "wd", "bd", "wqkv_hm", "bqkv_hm")]


class VitWeights(ctypes.Structure):
    """fo1_vit_weights_t"""
    _fields_ = [("depth", c_int32), ("hidden", c_int32), ("n_heads", c_int32), ("ff_padded", c_int32), ("k_in", c_int32),
                ("k_in_padded", c_int32), ("merge", c_int32), ("out_hidden", c_int32), ("n_fullatt", c_int32), ("fullatt", c_int32 * 8),
                ("patch_w", c_void_p), ("blocks", ctypes.POINTER(VitBlock)),
                ("ln_q", c_void_p), ("m0w", c_void_p), ("m0b", c_void_p), ("m2w", c_void_p), ("m2b", c_void_p)]


class VitPlan(ctypes.Structure):
    """fo1_vit_plan_t"""
    _fields_ = [("S", c_int32), ("plan_in", c_void_p), ("plan_raster", c_void_p), ("plan_tokens", c_void_p), ("cos", c_void_p), ("sin", c_void_p),
                ("items_win", c_void_p), ("n_items_win", c_int32), ("q_block_win", c_int32),
                ("items_full", c_void_p), ("n_items_full", c_int32), ("q_block_full", c_int32),
                ("flops_win", ctypes.c_double), ("flops_full", ctypes.c_double)]


class LlmLayer(ctypes.Structure):
    """fo1_llm_layer_t"""
    _fields_ = [(n, c_void_p) for n in ("ln1", "ln2", "wqkv", "bqkv", "wo", "wgu", "wdown")]


class LlmWeights(ctypes.Structure):
    """fo1_llm_weights_t"""
    _fields_ = [("n_layers", c_int32), ("hidden", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32), ("head_dim", c_int32),
                ("intermediate", c_int32), ("vocab", c_int32), ("rms_eps", c_float), ("layers", ctypes.POINTER(LlmLayer)),
                ("embed", c_void_p), ("final_norm", c_void_p), ("lm_head", c_void_p)]


class KvCache(ctypes.Structure):
    """fo1_kv_cache_t"""
    _fields_ = [("k", c_void_p), ("k_layer_stride", c_longlong), ("k_head_stride", c_longlong),
                ("vt", c_void_p), ("vt_layer_stride", c_longlong), ("vt_row_stride", c_longlong), ("capacity", c_int32)]


_HALF = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "an_w", "an_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "fn_w", "fn_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class DavitHalf(ctypes.Structure):
    """fo1_davit_half_t"""
    _fields_ = [(n, c_void_p) for n in _HALF]


class DavitBlock(ctypes.Structure):
    """fo1_davit_block_t"""
    _fields_ = [("spatial", DavitHalf), ("channel", DavitHalf)]


class DavitStage(ctypes.Structure):
    """fo1_davit_stage_t"""
    _fields_ = [(n, c_int32) for n in ("dim", "heads", "depth", "kernel", "stride", "pad", "prenorm", "K_padded")] + \
               [(n, c_void_p) for n in ("conv_w", "conv_b", "norm_w", "norm_b")] + [("blocks", ctypes.POINTER(DavitBlock))]


class DavitWeights(ctypes.Structure):
    """fo1_davit_weights_t"""
    _fields_ = [("n_stages", c_int32), ("window", c_int32), ("stages", DavitStage * 4)]


class DavitPlan(ctypes.Structure):
    """fo1_davit_plan_t"""
    _fields_ = [("H", c_int32), ("W", c_int32), ("batch", c_int32), ("items", c_void_p * 4), ("n_items", c_int32 * 4), ("q_block", c_int32 * 4)]


class FpnHead(ctypes.Structure):
    """fo1_fpn_head_t"""
    _fields_ = [(n, c_void_p) for n in ("w1", "n1_w", "n1_b", "w3", "n3_w", "n3_b")]


class FpnWeights(ctypes.Structure):
    """fo1_fpn_weights_t"""
    _fields_ = [(n, c_int32) for n in ("c_in", "c_up1", "c_up2", "c_out")] + \
               [(n, c_void_p) for n in ("t1a_w", "t1a_b", "t1_ln_w", "t1_ln_b", "t1b_w", "t1b_b", "t2_w", "t2_b")] + [("heads", FpnHead * 4)]


class ProjectorW(ctypes.Structure):
    """fo1_projector_t"""
    _fields_ = [("n_layers", c_int32), ("dims", c_int32 * 5), ("w", c_void_p * 4), ("b", c_void_p * 4)]


class ProfileRow(ctypes.Structure):
    """fo1_profile_row_t (include/fo1.h)."""
    _fields_ = [("name", ctypes.c_char * 48), ("calls", ctypes.c_int64), ("total_ms", ctypes.c_double),
                ("total_work", ctypes.c_double)]


# name -> (restype, argtypes); mirrors include/fo1.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "fo1_abi_version": (c_int, []),
    "fo1_last_error": (ctypes.c_char_p, []),
    "fo1_profile_enable": (c_int, [c_int]),
    "fo1_profile_read": (c_int, [ctypes.POINTER(ProfileRow), c_int, c_int]),
    "fo1_profile_stage": (c_int, [ctypes.c_char_p]),
    "fo1_vit_workspace_bytes": (c_size_t, [ctypes.POINTER(VitWeights), c_int]),
    "fo1_vit_forward": (c_int, [ctypes.POINTER(VitWeights), ctypes.POINTER(VitPlan), c_void_p, c_int, c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                c_size_t, c_void_p]),
    "fo1_llm_prefill_workspace_bytes": (c_size_t, [ctypes.POINTER(LlmWeights), c_int, c_int]),
    "fo1_llm_prefill": (c_int, [ctypes.POINTER(LlmWeights), ctypes.POINTER(KvCache), c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                c_int, ctypes.c_double, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fo1_llm_decode_workspace_bytes": (c_size_t, [ctypes.POINTER(LlmWeights), c_int, c_int]),
    "fo1_llm_decode_step": (c_int, [ctypes.POINTER(LlmWeights), ctypes.POINTER(KvCache), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fo1_zero_bytes": (c_int, [c_void_p, c_size_t, c_void_p]),
    "fo1_attention_window_bias_bf16": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong,
                                               c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               ctypes.c_double, c_void_p]),
    "fo1_swin_window_partition_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_swin_window_reverse_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_patch_merge_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_groupnorm_tokens_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fo1_groupnorm_tokens_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fo1_sine_embed_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fo1_box_refine_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_mask_rows_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_topk_workspace_bytes": (c_size_t, [c_int]),
    "fo1_topk_desc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fo1_gather_rows_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_msda_fused_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p]),
    "fo1_add_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_ms_deform_attn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                           c_int, c_void_p]),
    "fo1_davit_workspace_bytes": (c_size_t, [ctypes.POINTER(DavitWeights), ctypes.POINTER(DavitPlan)]),
    "fo1_davit_forward": (c_int, [ctypes.POINTER(DavitWeights), ctypes.POINTER(DavitPlan), c_void_p, c_int, ctypes.POINTER(c_void_p), c_void_p, c_size_t,
                                  c_void_p]),
    "fo1_simplefpn_workspace_bytes": (c_size_t, [ctypes.POINTER(FpnWeights), c_int, c_int, c_int]),
    "fo1_simplefpn_forward": (c_int, [ctypes.POINTER(FpnWeights), c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "fo1_projector_workspace_bytes": (c_size_t, [ctypes.POINTER(ProjectorW), c_int]),
    "fo1_projector_forward": (c_int, [ctypes.POINTER(ProjectorW), c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fo1_hfre_workspace_bytes": (c_size_t, [ctypes.POINTER(HfreSource), c_int, c_int]),
    "fo1_hfre_region_pool": (c_int, [ctypes.POINTER(HfreSource), c_int, c_void_p, c_int, c_void_p, c_float, c_float,
                                     c_int, c_int, c_float, c_float, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                     c_void_p]),
    "fo1_hfre_ex_workspace_bytes": (c_size_t, [ctypes.POINTER(HfreSource), c_int, c_int]),
    "fo1_hfre_region_pool_ex": (c_int, [ctypes.POINTER(HfreSource), c_int, c_void_p, c_int, c_void_p, c_float, c_float,
                                           c_int, c_int, c_float, c_float, c_void_p, c_int, c_int, ctypes.POINTER(HfreOpts), c_void_p, c_size_t,
                                           c_void_p]),
    "fo1_gemm_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                              c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_gemm_bf16_ws": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fo1_quantize_rows_e4m3": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_longlong, c_void_p, c_void_p]),
    "fo1_rmsnorm_quant_e4m3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_longlong, c_void_p, c_void_p]),
    "fo1_gemm_fp8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                             c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_rmsnorm_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "fo1_layernorm_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "fo1_swiglu_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_bias_act_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_argmax_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "fo1_attention_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fo1_attention_decode_bf16": (c_int, [c_void_p, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "fo1_rope_llm_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_longlong, c_int, c_void_p, c_void_p]),
    "fo1_decode_advance": (c_int, [c_void_p, c_void_p]),
    "fo1_decode_qkv_post_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p,
                                         c_longlong, c_void_p]),
    "fo1_pool_qkv_post_bf16": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p,
                                       c_longlong, c_void_p]),
    "fo1_pool_qkv_post_partials_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_longlong, c_void_p, c_longlong, c_void_p]),
    "fo1_gemm_bf16_partials": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fo1_splitk_residual_rmsnorm_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p,
                                                 c_int, c_void_p]),
    "fo1_gemv_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                              c_void_p, c_float, c_void_p]),
    "fo1_rope_vit_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "fo1_transpose_bf16": (c_int, [c_void_p, c_int, c_void_p, c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    "fo1_qkv_post_llm_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_longlong,
                                      c_void_p, c_longlong, c_int, c_void_p]),
    "fo1_patchify_u8_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_normalize_u8_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fo1_qkv_post_vit_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_longlong, c_void_p]),
    "fo1_attention_bf16": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p,
                                   c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_float, c_int, c_void_p, ctypes.c_double, c_void_p]),
    "fo1_gemm_takes_big_tile": (c_int, [c_int, c_int, c_int]),
    "fo1_conv3x3_gemm_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_layernorm_rows_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "fo1_qkv_proj_rope_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_longlong, c_void_p]),
    "fo1_attention_prefix_bf16": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p,
                                          c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_float, c_int, ctypes.c_double, c_void_p]),
    "fo1_dwconv3x3_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_dwconv3x3_ln_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int,
                                      c_int, c_void_p]),
    "fo1_im2col_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_window_partition_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_window_reverse_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_attention_windows_bf16": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong,
                                    c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_float, ctypes.c_double, c_void_p]),
    "fo1_window_attention_map_bf16": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_longlong, c_float, c_void_p]),
    "fo1_window_attention_map_var_bf16": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, c_longlong,
                                                 c_float, c_void_p]),
    "fo1_window_attention_bf16": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_longlong, c_float, c_void_p]),
    "fo1_channel_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fo1_channel_attention_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fo1_pixel_shuffle2_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_maxpool2_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_nchw_to_hwc8_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fo1_dwconv3x3_ln_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int,
                                          c_longlong, c_int, c_void_p]),
    "fo1_im2col_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_window_partition_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_void_p]),
    "fo1_window_reverse_add_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_void_p]),
    "fo1_channel_attention_var_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fo1_channel_attention_var_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_longlong, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fo1_pixel_shuffle2_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_void_p]),
    "fo1_maxpool2_var_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_void_p]),
    "fo1_gemv_batch_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong,
                                    c_void_p]),
    "fo1_attention_decode_batch_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fo1_attention_decode_batch_bf16": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p,
                                                c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "fo1_attention_decode_batch_partials_bf16": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_int,
                                                         c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "fo1_gemv_attn_combine_bf16": (c_int, [c_void_p, c_longlong, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                           c_void_p]),
    "fo1_decode_argmax_accept": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p]),
    "fo1_kv_relocate": (c_int, [c_void_p, c_void_p, c_longlong, c_longlong, c_longlong, c_longlong, c_void_p, c_void_p, c_longlong, c_longlong,
                                c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_gather_rows_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_void_p]),
}

# include/fo1_ab.h: A/B / ablation / determinism-pin switches — exported by libfo1hip_ab.so ONLY (FO1_AB=1), never by the product library
SIGNATURES_AB = {
    "fo1_hfre_set_pixel_budget": (c_int, [c_int]),
    "fo1_hfre_set_tuning": (c_int, [c_int, c_int, c_int, c_int]),
    "fo1_gemm_set_variant": (c_int, [c_int, c_int]),
    "fo1_gemm_set_splitk": (c_int, [c_int]),
    "fo1_gemm_set_gemv": (c_int, [c_int]),
    "fo1_gemm_set_group_m": (c_int, [c_int]),
    "fo1_gemm_set_big_schedule": (c_int, [c_int]),
    "fo1_gemm_set_debug": (c_int, [c_int]),
    "fo1_gemm_set_stamp_buffer": (c_int, [c_void_p]),
    "fo1_gemv_batch_set_rows_per_lane": (c_int, [c_int]),
    "fo1_gemv_batch_set_impl": (c_int, [c_int]),
    "fo1_attention_decode_set_impl": (c_int, [c_int]),
    "fo1_attention_decode_set_pool_chunk": (c_int, [c_int]),
    "fo1_attention_decode_set_small_chunk": (c_int, [c_int]),
    "fo1_dwconv_ln_set_form": (c_int, [c_int]),
    "fo1_channel_attention_set_impl": (c_int, [c_int]),
    # measured no-gain kernel forms and instruments (round 5: out of the product ABI)
    "fo1_gemm_profile_shapes": (c_int, [c_int]),
    "fo1_mfma_clock_probe": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fo1_traffic_probe": (c_int, [c_int, c_void_p, c_longlong, c_longlong, c_int, c_void_p, c_void_p]),
    "fo1_gemm_bf16_wtiled": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fo1_splitk_swiglu_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
}

_lib = None          # the ACTIVE library: every ops.* call goes through load()
_handles = {}        # path -> CDLL (both builds may be mapped in one process: separate copies of the code and of its static state)


class Fo1Error(RuntimeError):
    pass


def ab_build() -> bool:
    """True when the ACTIVE library is the test / bench build (libfo1hip_ab.so with include/fo1_ab.h's switches): a process started
    with FO1_AB=1 (the profiling scripts), or code inside `use_ab()` (the tests that pin tiles / routing)."""
    if _lib is not None:
        return _lib is _handles.get(LIB_PATH_AB)
    return os.environ.get("FO1_AB", "0") not in ("", "0")


def _open(path: str, ab: bool) -> ctypes.CDLL:
    lib = _handles.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise Fo1Error(
                f"{path} not found — build it with `python -m vlm_fo1_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no fallback path.")
        # torch (the device-memory / stream plumbing) must load ITS HIP runtime first so that
        # libfo1hip.so binds to the same libamdhip64/libhsa instance; loading ours first puts a
        # second runtime in the process and launches fail with "no ROCm-capable device".
        import torch  # noqa: F401
        lib = ctypes.CDLL(path)
        table = dict(SIGNATURES)
        if ab:
            table.update(SIGNATURES_AB)
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _handles[path] = lib
    return lib


def load() -> ctypes.CDLL:
    """The active library: the PRODUCT build unless the process was started with FO1_AB=1 or runs inside `use_ab()`."""
    global _lib
    if _lib is None:
        ab = ab_build()
        _lib = _open(LIB_PATH_AB if ab else LIB_PATH, ab)
    return _lib


def active_path() -> str:
    return load()._name


class use_ab:
    """Context manager: route every call through the test / bench build for the duration (tests that need include/fo1_ab.h's
    determinism pins or A/B kernels).  Everything launched inside — including hipGraphs captured inside — runs the AB build's
    kernels; everything outside runs the product library, which is what the parity tests therefore exercise by default.  Not
    thread-safe by design: it is a test-session switch, not a per-request one."""

    def __enter__(self):
        global _lib
        self._prev = load()
        _lib = _open(LIB_PATH_AB, True)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fo1_last_error().decode("utf-8", "replace")
        raise Fo1Error(f"{what} failed (rc={rc}): {msg}")


def current_stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def profile(on: bool) -> None:
    load().fo1_profile_enable(1 if on else 0)


def profile_stage(tag: str) -> None:
    """Label the kernel records since the previous call '<tag>|<kernel>' (no sync)."""
    check(load().fo1_profile_stage(tag.encode()), "fo1_profile_stage")


def profile_rows(reset: bool = True):
    """-> list of dict(name, calls, total_ms, total_work) since the last reset."""
    rows = (ProfileRow * 512)()
    n = load().fo1_profile_read(rows, 512, 1 if reset else 0)
    if n < 0:
        check(n, "fo1_profile_read")
    return [dict(name=rows[i].name.decode(), calls=rows[i].calls, total_ms=rows[i].total_ms,
                 total_work=rows[i].total_work) for i in range(min(n, 512))]
