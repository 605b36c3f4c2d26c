// stages.hip — stage-level C-ABI entries ("one extern "C" function per fused stage"): the launch sequences of the
// Qwen2.5-VL vision tower, the LLM prefill and the batched decode step as plain C calls over caller-owned device memory, so that a
// host that is not Python can drive the hot path.  Each entry issues the primitive launches the Python mirror issues
// (vlm_fo1_amd/vit.py QwenViT._forward, vlm_fo1_amd/llm.py QwenLLM._forward / prefill_packed, BatchDecoder._step_device) in the
// same order with the same arguments: results are bit-identical to that path (tests/test_stage_abi_gpu.py), and the calls are
// asynchronous on the caller's stream and safe to capture in a hipGraph (no allocation, no synchronisation, no memset node).
// Round 5's fused forms: fo1_llm_prefill takes the q/k/v epilogue (fo1_qkv_proj_rope_bf16) and the DaViT / SimpleFPN entries the
// implicit-GEMM convolution (index tables filled on the device) under the mirror's own rules, fo1_vit_forward the q/k/v epilogue when
// its weight table carries the head-major copy (wqkv_hm, ABI 7) — the same bits as the two-launch forms either way.
//
// Reference call sites replaced:
//   fo1_vit_forward      Qwen2_5_VisionTransformerPretrainedModel.forward  modeling_qwen2_5_vl.py:436-504 (blocks :306-357,
//                        merger :140-158) as driven by qwen2_5_vl_encoder.py:86-158,228-257 (window order, un-window, map capture)
//   fo1_llm_prefill      Qwen2_5_VLModel.forward over the spliced prompt   modeling_qwen2_5_vl.py:1126-1242 (decoder layer
//                        :1014-1095, attention :738-802) + last-row lm_head / greedy pick (omchat_qwen2_5_vl.py:143-155,
//                        modeling_qwen2_5_vl.py:1848-1860)
//   fo1_llm_decode_step  the 1-token fast path of the same code for B sequences
#include "common.h"

namespace {

struct Carver {   // bump allocator over the caller's workspace, 256-byte granules
    char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t n) : base((char*)p), cap(n) {}
    void* take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return base ? base + o : nullptr;
    }
};

constexpr size_t kGemmScratch = 64ull << 20;   // split-K partials: the size the Python mirror hands to fo1_gemm_bf16_ws

inline size_t bf16_rows(long long rows, long long cols) { return (size_t)rows * cols * 2; }

#define FO1_TRY(call)                 \
    do {                              \
        const int _rc = (call);       \
        if (_rc != FO1_OK) return _rc; \
    } while (0)

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// Vision tower
// ---------------------------------------------------------------------------------------------------------------------
static size_t vit_layout(const fo1_vit_weights_t* w, int S, int Sp, void* ws, size_t ws_bytes, void** xin, void** xa, void** xb, void** h,
                         void** qkv, void** att, void** a, void** vt, void** m0, void** m1, void** m2, void** gemm_ws) {
    Carver c(ws, ws_bytes);
    const int d = w->hidden, u = w->merge * w->merge;
    *xin = c.take(bf16_rows(S, w->k_in_padded));
    *xa = c.take(bf16_rows(S, d));
    *xb = c.take(bf16_rows(S, d));
    *h = c.take(bf16_rows(S, d));
    const bool hm = w->blocks && w->depth > 0 && w->blocks[0].wqkv_hm;      // head-major q/k/v copy present: rows of 256 columns per head
    *qkv = c.take(bf16_rows(S, hm && 256 * w->n_heads > 3 * d ? 256 * w->n_heads : 3 * d));
    *att = c.take(bf16_rows(S, d));
    *a = c.take(bf16_rows(S, w->ff_padded));
    *vt = c.take(bf16_rows(d, Sp));
    *m0 = c.take(bf16_rows(S, d));
    *m1 = c.take(bf16_rows(S / u, (long long)u * d));
    *m2 = c.take(bf16_rows(S / u, w->out_hidden));
    *gemm_ws = c.take(kGemmScratch);
    return c.off;
}

size_t fo1_vit_workspace_bytes(const fo1_vit_weights_t* w, int S) {
    if (!w || S <= 0) return 0;
    void* p[12];
    return vit_layout(w, S, (S + 63) / 64 * 64, nullptr, 0, &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6], &p[7], &p[8], &p[9], &p[10], &p[11]);
}

int fo1_vit_forward(const fo1_vit_weights_t* w, const fo1_vit_plan_t* g, const void* pixel_rows, int ld_pixels, void* tokens_out,
                    void* const* feature_maps_out, void* workspace, size_t workspace_bytes, void* stream) {
    FO1_CHECK_ARG(w && g && pixel_rows && tokens_out && w->blocks, "vit_forward: NULL argument");
    const int S = g->S, d = w->hidden, H = w->n_heads, hd = d / H, u = w->merge * w->merge;
    FO1_CHECK_ARG(S > 0 && S % u == 0 && d % H == 0, "vit_forward: S=%d must be a positive multiple of %d", S, u);
    FO1_CHECK_ARG(g->plan_in && g->plan_raster && g->plan_tokens && g->cos && g->sin && g->items_win && g->items_full,
                  "vit_forward: incomplete plan");
    const int Sp = (S + 63) / 64 * 64;
    void *xin, *xa, *xb, *h, *qkv, *att, *a, *vt, *m0, *m1, *m2, *gws;
    const size_t need = vit_layout(w, S, Sp, workspace, workspace_bytes, &xin, &xa, &xb, &h, &qkv, &att, &a, &vt, &m0, &m1, &m2, &gws);
    if (!workspace || workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "vit_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    // window re-order folded into the patch-embed input gather; K padded 1176 -> 1216 with zeros; V^T scratch zeroed (the
    // attention kernel may read up to 3 finite columns past a segment's end)
    // (a fill kernel, not hipMemsetAsync: memset nodes misbehaved on replay inside captured graphs on ROCm 7.2)
    FO1_TRY(fo1_zero_bytes(xin, bf16_rows(S, w->k_in_padded), stream));
    FO1_TRY(fo1_zero_bytes(vt, bf16_rows(d, Sp), stream));
    FO1_TRY(fo1_gather_rows_bf16(pixel_rows, ld_pixels, nullptr, 0, nullptr, 0, g->plan_in, xin, w->k_in_padded, S, w->k_in, stream));
    void* x = xa;
    void* xn = xb;
    FO1_TRY(fo1_gemm_bf16_ws(xin, w->k_in_padded, w->patch_w, w->k_in_padded, nullptr, nullptr, 0, x, d, S, d, w->k_in_padded, 0, 0, gws, kGemmScratch, stream));
    const float scale = (float)(1.0 / sqrt((double)hd));   // rounded once from double, like the Python mirror's argument
    int n_cap = 0;
    const bool fused_qkv = hd == 80 && d % 64 == 0 && fo1_gemm_takes_big_tile(S, 3 * d, d) == 1;
    for (int i = 0; i < w->depth; ++i) {
        const fo1_vit_block_t& b = w->blocks[i];
        bool full = false;
        for (int k = 0; k < w->n_fullatt; ++k) full = full || (w->fullatt[k] == i);
        FO1_TRY(fo1_rmsnorm_bf16(x, d, b.n1, h, d, S, d, 1e-6f, stream));
        if (fused_qkv && b.wqkv_hm && b.bqkv_hm) {
            // 2-D RoPE + V -> V^T in the q/k/v GEMM's epilogue (vit.py's rule: the same bits, one launch and one pass over [S, 3 d] less); q and k of
            // head j sit at columns 256 j and 256 j + 80 of the head-major rows
            FO1_TRY(fo1_qkv_proj_rope_bf16(h, d, b.wqkv_hm, d, b.bqkv_hm, qkv, 256 * H, S, 256 * H, d, 1, H, H, g->cos, g->sin, nullptr, 0, 0, vt, Sp, stream));
            if (!full && g->q_block_win == 0) {     // single-tile items (windows of <= 64 tokens): the pipelined kernel, the same bits
                FO1_TRY(fo1_attention_windows_bf16(qkv, 256 * H, 256, (const uint16_t*)qkv + hd, 256 * H, 256, vt, Sp, att, d, hd, S, g->items_win, g->n_items_win,
                                                   H, H, hd, scale, g->flops_win, stream));
            } else {
                FO1_TRY(fo1_attention_bf16(qkv, 256 * H, 256, (const uint16_t*)qkv + hd, 256 * H, 256, vt, Sp, att, d, hd,
                                           full ? g->items_full : g->items_win, full ? g->n_items_full : g->n_items_win,
                                           full ? g->q_block_full : g->q_block_win, H, H, hd, scale, 0, nullptr,
                                           full ? g->flops_full : g->flops_win, stream));
            }
        } else {
            FO1_TRY(fo1_gemm_bf16_ws(h, d, b.wqkv, d, b.bqkv, nullptr, 0, qkv, 3 * d, S, 3 * d, d, 0, 0, gws, kGemmScratch, stream));
            FO1_TRY(fo1_qkv_post_vit_bf16(qkv, 3 * d, H, hd, g->cos, g->sin, S, vt, Sp, stream));   // 2-D RoPE on q/k + V -> V^T
            if (!full && g->q_block_win == 0) {
                FO1_TRY(fo1_attention_windows_bf16(qkv, 3 * d, hd, (const uint16_t*)qkv + d, 3 * d, hd, vt, Sp, att, d, hd, S, g->items_win, g->n_items_win, H, H, hd,
                                                   scale, g->flops_win, stream));
            } else {
                FO1_TRY(fo1_attention_bf16(qkv, 3 * d, hd, (const uint16_t*)qkv + d, 3 * d, hd, vt, Sp, att, d, hd,
                                           full ? g->items_full : g->items_win, full ? g->n_items_full : g->n_items_win,
                                           full ? g->q_block_full : g->q_block_win, H, H, hd, scale, 0, nullptr,
                                           full ? g->flops_full : g->flops_win, stream));
            }
        }
        FO1_TRY(fo1_gemm_bf16_ws(att, d, b.wo, d, b.bo, x, d, xn, d, S, d, d, 0, 0, gws, kGemmScratch, stream));
        { void* t = x; x = xn; xn = t; }
        FO1_TRY(fo1_rmsnorm_bf16(x, d, b.n2, h, d, S, d, 1e-6f, stream));
        FO1_TRY(fo1_gemm_bf16_ws(h, d, b.wgu, d, b.bgu, nullptr, 0, a, w->ff_padded, S, 2 * w->ff_padded, d, 3, 0, gws, kGemmScratch, stream));
        FO1_TRY(fo1_gemm_bf16_ws(a, w->ff_padded, b.wd, w->ff_padded, b.bd, x, d, xn, d, S, d, w->ff_padded, 0, 0, gws, kGemmScratch, stream));
        { void* t = x; x = xn; xn = t; }
        if (full) {
            // un-window: the raster token-major map of this full-attention block (qwen2_5_vl_encoder.py:37-80)
            void* dst = feature_maps_out ? feature_maps_out[n_cap] : nullptr;
            if (dst) FO1_TRY(fo1_gather_rows_bf16(x, d, nullptr, 0, nullptr, 0, g->plan_raster, dst, d, S, d, stream));
            ++n_cap;
        }
    }
    // merger: RMSNorm -> [S/4, 4d] -> Linear + GELU -> Linear, then raster-merged token order
    FO1_TRY(fo1_rmsnorm_bf16(x, d, w->ln_q, m0, d, S, d, 1e-6f, stream));
    FO1_TRY(fo1_gemm_bf16_ws(m0, u * d, w->m0w, u * d, w->m0b, nullptr, 0, m1, u * d, S / u, u * d, u * d, 1, 0, gws, kGemmScratch, stream));
    FO1_TRY(fo1_gemm_bf16_ws(m1, u * d, w->m2w, u * d, w->m2b, nullptr, 0, m2, w->out_hidden, S / u, w->out_hidden, u * d, 0, 0, gws, kGemmScratch, stream));
    FO1_TRY(fo1_gather_rows_bf16(m2, w->out_hidden, nullptr, 0, nullptr, 0, g->plan_tokens, tokens_out, w->out_hidden, S / u, w->out_hidden, stream));
    return FO1_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LLM prefill
// ---------------------------------------------------------------------------------------------------------------------
static size_t llm_prefill_layout(const fo1_llm_weights_t* w, int R, int n_seq, void* ws, size_t ws_bytes, void** xa, void** xb, void** h,
                                 void** qkv, void** att, void** a, void** last_rows, void** argmax_sc, void** gemm_ws) {
    Carver c(ws, ws_bytes);
    const int d = w->hidden, qd = (w->n_heads + 2 * w->n_kv_heads) * w->head_dim;
    *xa = c.take(bf16_rows(R, d));
    *xb = c.take(bf16_rows(R, d));
    *h = c.take(bf16_rows(R, d));
    *qkv = c.take(bf16_rows(R, qd));
    *att = c.take(bf16_rows(R, (long long)w->n_heads * w->head_dim));
    *a = c.take(bf16_rows(R, w->intermediate));
    *last_rows = c.take(bf16_rows(n_seq, d));
    *argmax_sc = c.take(4096);
    *gemm_ws = c.take(kGemmScratch);
    return c.off;
}

size_t fo1_llm_prefill_workspace_bytes(const fo1_llm_weights_t* w, int rows, int n_seq) {
    if (!w || rows <= 0 || n_seq <= 0) return 0;
    void* p[9];
    return llm_prefill_layout(w, rows, n_seq, nullptr, 0, &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6], &p[7], &p[8]);
}

int fo1_llm_prefill(const fo1_llm_weights_t* w, const fo1_kv_cache_t* kv, const void* embeds, int ld_embeds, const void* cos, const void* sin,
                    int rows, int pos0, const int32_t* items, int n_items, int q_block, double attn_flops, const int32_t* last_plan,
                    int n_seq, void* hidden_out, void* last_hidden, void* logits, int32_t* next_ids, void* workspace, size_t workspace_bytes,
                    void* stream) {
    FO1_CHECK_ARG(w && kv && embeds && cos && sin && items && last_plan && last_hidden && logits && next_ids && w->layers, "llm_prefill: NULL argument");
    const int R = rows, d = w->hidden, H = w->n_heads, KV = w->n_kv_heads, HD = w->head_dim, I = w->intermediate;
    FO1_CHECK_ARG(R > 0 && n_seq > 0 && pos0 >= 0 && pos0 + R <= kv->capacity, "llm_prefill: rows %d at %d exceed the KV cache (%d rows)", R, pos0, kv->capacity);
    void *xa, *xb, *h, *qkv, *att, *a, *last_rows, *asc, *gws;
    const size_t need = llm_prefill_layout(w, R, n_seq, workspace, workspace_bytes, &xa, &xb, &h, &qkv, &att, &a, &last_rows, &asc, &gws);
    if (!workspace || workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "llm_prefill: workspace %zu B < required %zu B", workspace_bytes, need);
    const int qd = (H + 2 * KV) * HD;
    const float scale = (float)(1.0 / sqrt((double)HD));
    const void* x = embeds;
    int ldx = ld_embeds;
    // the Python mirror's rule (llm.py:_forward): where the projection runs on the 256 x 256 GEMM kernel anyway, mRoPE + K append + V^T ride in
    // its epilogue (fo1_qkv_proj_rope_bf16: the same bits as fo1_gemm_bf16 + fo1_qkv_post_llm_bf16, one launch and one pass over [R, qd] less)
    const bool fused_qkv = HD == 128 && pos0 % 8 == 0 && kv->vt_row_stride % 8 == 0 && kv->k_head_stride % 8 == 0 && qd % 256 == 0 &&
                           ((uintptr_t)cos & 15) == 0 && ((uintptr_t)sin & 15) == 0 && fo1_gemm_takes_big_tile(R, qd, d) == 1;
    for (int li = 0; li < w->n_layers; ++li) {
        const fo1_llm_layer_t& L = w->layers[li];
        uint16_t* kc = (uint16_t*)kv->k + (long long)li * kv->k_layer_stride;
        uint16_t* vtc = (uint16_t*)kv->vt + (long long)li * kv->vt_layer_stride;
        FO1_TRY(fo1_rmsnorm_bf16(x, ldx, L.ln1, h, d, R, d, w->rms_eps, stream));
        if (fused_qkv) {
            FO1_TRY(fo1_qkv_proj_rope_bf16(h, d, L.wqkv, d, L.bqkv, qkv, qd, R, qd, d, 0, H, KV, cos, sin, kc, kv->k_head_stride, pos0, vtc, kv->vt_row_stride,
                                           stream));
        } else {
            FO1_TRY(fo1_gemm_bf16_ws(h, d, L.wqkv, d, L.bqkv, nullptr, 0, qkv, qd, R, qd, d, 0, 0, gws, kGemmScratch, stream));
            // mRoPE on q/k + K append + V^T, one launch
            FO1_TRY(fo1_qkv_post_llm_bf16(qkv, qd, H, KV, HD, cos, sin, R, kc, kv->k_head_stride, vtc, kv->vt_row_stride, pos0, stream));
        }
        // causal attention of the packed segments against the cache
        FO1_TRY(fo1_attention_bf16((const uint16_t*)qkv - (long long)pos0 * qd, qd, HD, kc, HD, kv->k_head_stride, vtc, kv->vt_row_stride,
                                   (uint16_t*)att - (long long)pos0 * H * HD, (long long)H * HD, HD, items, n_items, q_block, H, KV, HD, scale, 1,
                                   nullptr, attn_flops, stream));
        // residual stream: x -> xa (after attention) -> xb (after the MLP; the old x is dead by then, so xb may be what held it)
        FO1_TRY(fo1_gemm_bf16_ws(att, H * HD, L.wo, H * HD, nullptr, x, ldx, xa, d, R, d, H * HD, 0, 0, gws, kGemmScratch, stream));
        FO1_TRY(fo1_rmsnorm_bf16(xa, d, L.ln2, h, d, R, d, w->rms_eps, stream));
        FO1_TRY(fo1_gemm_bf16_ws(h, d, L.wgu, d, nullptr, nullptr, 0, a, I, R, 2 * I, d, 3, 0, gws, kGemmScratch, stream));
        void* dst = (li + 1 == w->n_layers && hidden_out) ? hidden_out : xb;
        FO1_TRY(fo1_gemm_bf16_ws(a, I, L.wdown, I, nullptr, xa, d, dst, d, R, d, I, 0, 0, gws, kGemmScratch, stream));
        x = dst;
        ldx = d;
    }
    // last-row head: gather each sequence's final row, final RMSNorm, lm_head, greedy pick
    FO1_TRY(fo1_gather_rows_bf16(x, ldx, nullptr, 0, nullptr, 0, last_plan, last_rows, d, n_seq, d, stream));
    FO1_TRY(fo1_rmsnorm_bf16(last_rows, d, w->final_norm, last_hidden, d, n_seq, d, w->rms_eps, stream));
    FO1_TRY(fo1_gemm_bf16_ws(last_hidden, d, w->lm_head, d, nullptr, nullptr, 0, logits, w->vocab, n_seq, w->vocab, d, 0, 0, gws, kGemmScratch, stream));
    for (int b = 0; b < n_seq; ++b)
        FO1_TRY(fo1_argmax_bf16((const uint16_t*)logits + (long long)b * w->vocab, w->vocab, next_ids + b, asc, stream));
    return FO1_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched decode step
// ---------------------------------------------------------------------------------------------------------------------
static size_t llm_decode_layout(const fo1_llm_weights_t* w, int B, int slot_rows, void* ws, size_t ws_bytes, void** xa, void** xb,
                                void** q, void** att, void** a, void** attn_ws, size_t* attn_bytes, void** argmax_sc) {
    Carver c(ws, ws_bytes);
    const int d = w->hidden;
    *xa = c.take(bf16_rows(B, d));
    *xb = c.take(bf16_rows(B, d));
    *q = c.take(bf16_rows(B, (long long)w->n_heads * w->head_dim));
    *att = c.take(bf16_rows(B, (long long)w->n_heads * w->head_dim));
    *a = c.take(bf16_rows(B, w->intermediate));
    *attn_bytes = fo1_attention_decode_batch_workspace_bytes(slot_rows, w->n_kv_heads, w->head_dim, B);
    *attn_ws = c.take(*attn_bytes);
    *argmax_sc = c.take((size_t)2 * 128 * B * 4);
    return c.off;
}

size_t fo1_llm_decode_workspace_bytes(const fo1_llm_weights_t* w, int batch, int slot_rows) {
    if (!w || batch <= 0 || slot_rows <= 0) return 0;
    void* p[7];
    size_t ab;
    return llm_decode_layout(w, batch, slot_rows, nullptr, 0, &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &ab, &p[6]);
}

int fo1_llm_decode_step(const fo1_llm_weights_t* w, const fo1_kv_cache_t* slots, const void* rope_cos, const void* rope_sin, int32_t* state,
                        int32_t* plan, int32_t* ids_out, int ids_ld, const int32_t* stop_ids, int n_stop, int32_t* done, int batch,
                        int slot_rows, int max_kv_len, void* logits, void* workspace, size_t workspace_bytes, void* stream) {
    FO1_CHECK_ARG(w && slots && rope_cos && rope_sin && state && plan && ids_out && done && logits && w->layers && w->embed, "llm_decode_step: NULL argument");
    const int B = batch, d = w->hidden, H = w->n_heads, KV = w->n_kv_heads, HD = w->head_dim, I = w->intermediate;
    FO1_CHECK_ARG(B >= 1 && B <= 32 && HD == 128, "llm_decode_step: batch %d (1..32), head_dim %d (128)", B, HD);
    FO1_CHECK_ARG(max_kv_len >= 1 && max_kv_len <= slot_rows, "llm_decode_step: max_kv_len %d outside [1, slot_rows = %d]", max_kv_len, slot_rows);
    void *xa, *xb, *q, *att, *a, *aws, *asc;
    size_t abytes;
    const size_t need = llm_decode_layout(w, B, slot_rows, workspace, workspace_bytes, &xa, &xb, &q, &att, &a, &aws, &abytes, &asc);
    if (!workspace || workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "llm_decode_step: workspace %zu B < required %zu B", workspace_bytes, need);
    const int qd = (H + 2 * KV) * HD;
    const float scale = (float)(1.0 / sqrt((double)HD));
    // embedding rows of the tokens accepted by the previous step (plan = {0, token id} per sequence)
    FO1_TRY(fo1_gather_rows_bf16(w->embed, d, nullptr, 0, nullptr, 0, plan, xa, d, B, d, stream));
    void* x = xa;      // residual stream before attention / after the MLP
    void* y = xb;      // ... after attention
    for (int li = 0; li < w->n_layers; ++li) {
        const fo1_llm_layer_t& L = w->layers[li];
        uint16_t* kc = (uint16_t*)slots->k + (long long)li * slots->k_layer_stride;
        uint16_t* vtc = (uint16_t*)slots->vt + (long long)li * slots->vt_layer_stride;
        // input_layernorm + QKV + bias + mRoPE + K row / V^T column append, one launch; q rows out
        FO1_TRY(fo1_gemv_batch_bf16(x, d, L.wqkv, d, L.bqkv, nullptr, 0, q, H * HD, B, qd, d, 2, L.ln1, w->rms_eps, H, KV, rope_cos, rope_sin, state, kc,
                                    slots->k_head_stride, vtc, slots->vt_row_stride, stream));
        if (B <= 2 && H * HD <= 2048 && d <= 4096) {
            // one or two sequences: the o-projection sums the split-KV partials in its prologue (fo1_gemv_attn_combine_bf16) — no combine launch,
            // same bits (the Python mirror, llm.BatchDecoder, takes the same route)
            int chunk = 0;
            long long pstride = 0;
            FO1_TRY(fo1_attention_decode_batch_partials_bf16(q, (long long)H * HD, kc, HD, slots->k_head_stride, vtc, slots->vt_row_stride, state, B, max_kv_len, H, KV, HD,
                                                             scale, aws, abytes, &chunk, &pstride, stream));
            FO1_TRY(fo1_gemv_attn_combine_bf16((const float*)aws, pstride, state, chunk, H, KV, L.wo, H * HD, x, d, y, d, B, d, stream));
        } else {
            FO1_TRY(fo1_attention_decode_batch_bf16(q, (long long)H * HD, kc, HD, slots->k_head_stride, vtc, slots->vt_row_stride, att, (long long)H * HD, state, B,
                                                    max_kv_len, H, KV, HD, scale, aws, abytes, stream));
            FO1_TRY(fo1_gemv_batch_bf16(att, H * HD, L.wo, H * HD, nullptr, x, d, y, d, B, d, H * HD, 0, nullptr, 0.f, 0, 0, nullptr, nullptr, nullptr, nullptr, 0,
                                        nullptr, 0, stream));
        }
        // post_attention_layernorm + gate/up + SwiGLU, one launch; then down + residual
        FO1_TRY(fo1_gemv_batch_bf16(y, d, L.wgu, d, nullptr, nullptr, 0, a, I, B, 2 * I, d, 1, L.ln2, w->rms_eps, 0, 0, nullptr, nullptr, nullptr, nullptr, 0,
                                    nullptr, 0, stream));
        FO1_TRY(fo1_gemv_batch_bf16(a, I, L.wdown, I, nullptr, y, d, x, d, B, d, I, 0, nullptr, 0.f, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0,
                                    stream));
    }
    // final norm + lm_head, greedy pick and on-device accept (stop rule, state advance, next gather plan)
    FO1_TRY(fo1_gemv_batch_bf16(x, d, w->lm_head, d, nullptr, nullptr, 0, logits, w->vocab, B, w->vocab, d, 0, w->final_norm, w->rms_eps, 0, 0, nullptr, nullptr,
                                nullptr, nullptr, 0, nullptr, 0, stream));
    FO1_TRY(fo1_decode_argmax_accept(logits, w->vocab, w->vocab, B, nullptr, state, plan, ids_out, ids_ld, n_stop ? stop_ids : nullptr, n_stop, done, asc, stream));
    return FO1_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DaViT-L aux tower, SimpleFPN, projector MLPs.  One routine serves both the workspace query (dry run: base == NULL, no
// launches) and the call itself, so the two cannot disagree.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Arena {   // stack allocator over the workspace; dry run when base is NULL (tracks the high-water mark)
    char* base;
    size_t off = 0, peak = 0;
    explicit Arena(void* p) : base((char*)p) {}
    void* take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        if (off > peak) peak = off;
        return base ? base + o : (void*)(uintptr_t)(o + 256);   // non-NULL placeholder in a dry run (never dereferenced)
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    bool dry() const { return base == nullptr; }
};
#define FO1_RUN(call)                            \
    do {                                         \
        if (!A.dry()) {                          \
            const int _rc = (call);              \
            if (_rc != FO1_OK) return _rc;       \
        }                                        \
    } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// Index tables of an implicit-GEMM 3x3 / pad 1 convolution (fo1_conv3x3_gemm_bf16) over B images of one size — what ops.Conv3x3Plan builds on the
// host for the Python mirror, the same values: rowmap[b H W + y W + x] = padded row of input pixel (b, y, x) in the zero-framed map (row pitch
// W + 2 pixels, (H + 2)(W + 2) rows per image), a_rows[b Ho Wo + oy Wo + ox] = BYTE offset of output pixel (b, oy, ox)'s top-left tap.
__global__ __launch_bounds__(256) void conv_plan_fill_kernel(int32_t* rowmap, uint32_t* a_rows, int H, int W, int B, int Ho, int Wo, int stride, int cin) {
    const long long Wp = W + 2, blk = (long long)(H + 2) * Wp;
    const long long n_in = (long long)B * H * W, n_out = (long long)B * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in + n_out; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_in) {
            const long long b = i / ((long long)H * W), r = i - b * H * W, y = r / W, x = r - y * W;
            rowmap[i] = (int32_t)(b * blk + (y + 1) * Wp + (x + 1));
        } else {
            const long long j = i - n_in, b = j / ((long long)Ho * Wo), r = j - b * Ho * Wo, oy = r / Wo, ox = r - oy * Wo;
            a_rows[j] = (uint32_t)((b * blk + oy * stride * Wp + ox * stride) * (long long)(cin * 2));
        }
    }
}

// the Python mirror's rule (ops.conv3x3_implicit_ok + Conv3x3Plan's 32-bit offsets): a 3x3 / pad 1 convolution over >= 64 power-of-two channels
// whose im2col GEMM would run on the 256 x 256 kernel anyway takes the implicit form — the same bits without the [M, 9 Cin] column matrix
inline bool conv_implicit_ok(int M_out, int cout, int cin, int k, int pad, int K_padded, int B, int H, int W) {
    return k == 3 && pad == 1 && cin >= 64 && (cin & (cin - 1)) == 0 && cout % 8 == 0 && K_padded == 9 * cin &&
           (long long)B * (H + 2) * (W + 2) * cin * 2 < (1ll << 32) && fo1_gemm_takes_big_tile(M_out, cout, 9 * cin) == 1;
}

// [LayerNorm ->] 3x3 convolution as an implicit GEMM: the norm writes the zero-framed map, the GEMM gathers its nine taps from it.
// src rows [B H W, cin] -> dst rows [B Ho Wo, cout].
int conv3x3_implicit(Arena& A, const void* src, const void* ln_w, const void* ln_b, float eps, const void* conv_w, const void* conv_b, void* dst,
                     int H, int W, int B, int stride, int cin, int cout, void* stream) {
    const size_t m = A.mark();
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1, Wp = W + 2;
    const long long n_in = (long long)B * H * W, n_out = (long long)B * Ho * Wo, pad_rows = (long long)B * (H + 2) * Wp;
    void* xpad = A.take(bf16_rows(pad_rows, cin));
    int32_t* rowmap = (int32_t*)A.take((size_t)n_in * 4);
    uint32_t* a_rows = (uint32_t*)A.take((size_t)n_out * 4);
    FO1_RUN(fo1_zero_bytes(xpad, bf16_rows(pad_rows, cin), stream));
    if (!A.dry()) {
        const long long tot = n_in + n_out;
        const int grid = (int)(tot / 256 + 1 < 2048 ? tot / 256 + 1 : 2048);
        FO1_LAUNCH("conv_plan_fill", (double)tot * 4.0, conv_plan_fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rowmap, a_rows, H, W, B, Ho, Wo,
                   stride, cin);
    }
    FO1_RUN(fo1_layernorm_rows_bf16(src, cin, ln_w, ln_b, xpad, cin, rowmap, (int)n_in, cin, eps, stream));
    FO1_RUN(fo1_conv3x3_gemm_bf16(xpad, a_rows, Wp, cin, conv_w, 9 * cin, conv_b, dst, cout, (int)n_out, cout, 0, stream));
    A.release(m);
    return FO1_OK;
}

// conv2 (depthwise 3x3 + residual) -> LayerNorm -> MLP (+ residual) -> dst      (modeling_davit.py:29-48,72-99,51-69)
int davit_conv_ffn(Arena& A, const fo1_davit_half_t& d, const void* x, void* tmp_x, void* dst, int n, int H, int W, int C, int B, void* gws, void* stream) {
    const size_t m = A.mark();
    void* h = A.take(bf16_rows(n, C));
    void* h1 = A.take(bf16_rows(n, 4 * C));
    FO1_RUN(fo1_dwconv3x3_ln_bf16(x, d.conv2_w, d.conv2_b, tmp_x, d.fn_w, d.fn_b, 1e-5f, h, H, W, C, B, stream));
    FO1_RUN(fo1_gemm_bf16_ws(h, C, d.fc1_w, C, d.fc1_b, nullptr, 0, h1, 4 * C, n, 4 * C, C, 1, 0, gws, kGemmScratch, stream));
    FO1_RUN(fo1_gemm_bf16_ws(h1, 4 * C, d.fc2_w, 4 * C, d.fc2_b, tmp_x, C, dst, C, n, C, 4 * C, 0, 0, gws, kGemmScratch, stream));
    A.release(m);
    return FO1_OK;
}

int davit_run(Arena& A, const fo1_davit_weights_t* w, const fo1_davit_plan_t* pl, const void* img, int img_is_f32, void* const* outs, void* stream) {
    const int B = pl->batch, ws = w->window;
    int H = pl->H, W = pl->W;
    void* gws = A.take(kGemmScratch);
    const void* prev = nullptr;   // previous stage's map (rows [B*H*W, Cprev])
    int Cprev = 8;
    {
        void* x0 = A.take(bf16_rows((long long)B * H * W, 8));
        FO1_RUN(fo1_nchw_to_hwc8_bf16(img, img_is_f32, x0, H, W, B, stream));
        prev = x0;
    }
    for (int i = 0; i < w->n_stages; ++i) {
        const fo1_davit_stage_t& sg = w->stages[i];
        const int C = sg.dim, heads = sg.heads, k = sg.kernel, st = sg.stride, pd = sg.pad;
        const int Ho = (H + 2 * pd - k) / st + 1, Wo = (W + 2 * pd - k) / st + 1;
        const int n_prev = B * H * W, n = B * Ho * Wo;
        const size_t stage_mark = A.mark();
        void* XA = A.take(bf16_rows(n, C));
        void* XB = A.take(bf16_rows(n, C));
        if (i > 0 && sg.prenorm && conv_implicit_ok(n, C, Cprev, k, pd, sg.K_padded, B, H, W)) {
            // pre-norm ConvEmbed as an implicit GEMM (as davit.py does for the same shapes: same bits, no column matrix)
            const int rc = conv3x3_implicit(A, prev, sg.norm_w, sg.norm_b, 1e-5f, sg.conv_w, sg.conv_b, XA, H, W, B, st, Cprev, C, stream);
            if (rc != FO1_OK) return rc;
        } else {   // ConvEmbed (modeling_davit.py:102-148): [pre-norm] -> conv as im2col + GEMM -> [post-norm]
            const size_t m = A.mark();
            const void* src = prev;
            if (i > 0 && sg.prenorm) {
                void* ln = A.take(bf16_rows(n_prev, Cprev));
                FO1_RUN(fo1_layernorm_bf16(prev, Cprev, sg.norm_w, sg.norm_b, ln, Cprev, n_prev, Cprev, 1e-5f, stream));
                src = ln;
            }
            void* col = A.take(bf16_rows(n, sg.K_padded));
            if (sg.K_padded != k * k * Cprev) FO1_RUN(fo1_zero_bytes(col, bf16_rows(n, sg.K_padded), stream));
            FO1_RUN(fo1_im2col_bf16(src, col, H, W, Cprev, k, k, st, pd, sg.K_padded, B, stream));
            if (i == 0 || !sg.prenorm) {
                FO1_RUN(fo1_gemm_bf16_ws(col, sg.K_padded, sg.conv_w, sg.K_padded, sg.conv_b, nullptr, 0, XB, C, n, C, sg.K_padded, 0, 0, gws, kGemmScratch, stream));
                FO1_RUN(fo1_layernorm_bf16(XB, C, sg.norm_w, sg.norm_b, XA, C, n, C, 1e-5f, stream));
            } else {
                FO1_RUN(fo1_gemm_bf16_ws(col, sg.K_padded, sg.conv_w, sg.K_padded, sg.conv_b, nullptr, 0, XA, C, n, C, sg.K_padded, 0, 0, gws, kGemmScratch, stream));
            }
            A.release(m);
        }
        H = Ho; W = Wo;
        // window geometry of this stage; V^T scratch zeroed once per stage (pad columns are never written afterwards)
        const int nWin = ((H + ws - 1) / ws) * ((W + ws - 1) / ws), nw = B * nWin * ws * ws, nw_pad = rup(nw, 64);
        void* vt = A.take(bf16_rows(C, nw_pad));
        FO1_RUN(fo1_zero_bytes(vt, bf16_rows(C, nw_pad), stream));
        const int hd = C / heads;
        for (int j = 0; j < sg.depth; ++j) {
            const fo1_davit_block_t& blk = sg.blocks[j];
            const bool last = (j + 1 == sg.depth);
            {   // SpatialBlock (:284-330): conv1 -> window attention (+ residual) -> conv2 -> FFN
                const fo1_davit_half_t& d = blk.spatial;
                const size_t m = A.mark();
                void* h = A.take(bf16_rows(n, C));
                void* hw = A.take(bf16_rows(nw, C));
                void* qkv = A.take(bf16_rows(nw, 3 * C));
                void* att = A.take(bf16_rows(nw, C));
                void* y = A.take(bf16_rows(nw, C));
                FO1_RUN(fo1_dwconv3x3_ln_bf16(XA, d.conv1_w, d.conv1_b, XB, d.an_w, d.an_b, 1e-5f, h, H, W, C, B, stream));
                if (hd == 32 && ws == 12 && d.qkv_b) {
                    // davit.py's rule: no partition / padded GEMM rows / reverse — the attention finds the windows' tokens among the pixel rows, padded
                    // tokens read the q/k/v bias row, the residual rides in the proj GEMM's epilogue (qkv / att: the first n rows of the buffers)
                    FO1_RUN(fo1_gemm_bf16_ws(h, C, d.qkv_w, C, d.qkv_b, nullptr, 0, qkv, 3 * C, n, 3 * C, C, 0, 0, gws, kGemmScratch, stream));
                    FO1_RUN(fo1_window_attention_map_bf16(qkv, 3 * C, C, heads, ws, H, W, B, d.qkv_b, att, C, (float)pow((double)hd, -0.5), stream));
                    FO1_RUN(fo1_gemm_bf16_ws(att, C, d.proj_w, C, d.proj_b, XB, C, XA, C, n, C, C, 0, 0, gws, kGemmScratch, stream));
                } else {
                    FO1_RUN(fo1_window_partition_bf16(h, hw, H, W, C, ws, B, stream));   // zero-padded AFTER the norm (:248-251)
                    FO1_RUN(fo1_gemm_bf16_ws(hw, C, d.qkv_w, C, d.qkv_b, nullptr, 0, qkv, 3 * C, nw, 3 * C, C, 0, 0, gws, kGemmScratch, stream));
                    if (hd == 32 && ws * ws <= 160) {      // every DaViT stage: the window kernel on the q/k/v rows themselves (no V^T copy)
                        FO1_RUN(fo1_window_attention_bf16(qkv, 3 * C, C, heads, ws * ws, nw / (ws * ws), att, C, (float)pow((double)hd, -0.5), stream));
                    } else {
                        FO1_RUN(fo1_transpose_bf16((const uint16_t*)qkv + 2 * C, 3 * C, vt, nw_pad, 0, nullptr, nw, C, stream));
                        FO1_RUN(fo1_attention_bf16(qkv, 3 * C, hd, (const uint16_t*)qkv + C, 3 * C, hd, vt, nw_pad, att, C, hd, pl->items[i], pl->n_items[i],
                                                   pl->q_block[i], heads, heads, hd, (float)pow((double)hd, -0.5), 0, nullptr, 4.0 * C * nw * ws * ws, stream));
                    }
                    FO1_RUN(fo1_gemm_bf16_ws(att, C, d.proj_w, C, d.proj_b, nullptr, 0, y, C, nw, C, C, 0, 0, gws, kGemmScratch, stream));
                    FO1_RUN(fo1_window_reverse_add_bf16(y, XB, XA, H, W, C, ws, B, stream));
                }
                A.release(m);
                const int rc = davit_conv_ffn(A, d, XA, XB, XA, n, H, W, C, B, gws, stream);
                if (rc != FO1_OK) return rc;
            }
            {   // ChannelBlock (:175-205): conv1 -> channel attention (+ residual) -> conv2 -> FFN
                const fo1_davit_half_t& d = blk.channel;
                const size_t m = A.mark();
                void* h = A.take(bf16_rows(n, C));
                void* qkv = A.take(bf16_rows(n, 3 * C));
                void* a = A.take(bf16_rows(n, C));
                const size_t cab = fo1_channel_attention_workspace_bytes(n / B, C, B);
                void* caw = A.take(cab);
                FO1_RUN(fo1_dwconv3x3_ln_bf16(XA, d.conv1_w, d.conv1_b, XB, d.an_w, d.an_b, 1e-5f, h, H, W, C, B, stream));
                FO1_RUN(fo1_gemm_bf16_ws(h, C, d.qkv_w, C, d.qkv_b, nullptr, 0, qkv, 3 * C, n, 3 * C, C, 0, 0, gws, kGemmScratch, stream));
                FO1_RUN(fo1_channel_attention_bf16(qkv, 3 * C, n / B, C, a, C, B, caw, cab, stream));
                FO1_RUN(fo1_gemm_bf16_ws(a, C, d.proj_w, C, d.proj_b, XB, C, XA, C, n, C, C, 0, 0, gws, kGemmScratch, stream));
                A.release(m);
                const int rc = davit_conv_ffn(A, d, XA, XB, last ? outs[i] : XA, n, H, W, C, B, gws, stream);
                if (rc != FO1_OK) return rc;
            }
        }
        prev = outs[i];
        Cprev = C;
        A.release(stage_mark);
    }
    return FO1_OK;
}

// one pyramid head: 1x1 conv -> LN -> 3x3 conv (im2col + GEMM) -> LN       (simple_fpn.py:165-175)
int fpn_head(Arena& A, const fo1_fpn_head_t& hd, const void* x, int Cin, void* dst, int H, int W, int B, int Cout, void* gws, void* stream) {
    const size_t m = A.mark();
    const int n = B * H * W;
    void* y = A.take(bf16_rows(n, Cout));
    void* y2 = A.take(bf16_rows(n, Cout));
    FO1_RUN(fo1_gemm_bf16_ws(x, Cin, hd.w1, Cin, nullptr, nullptr, 0, y, Cout, n, Cout, Cin, 0, 0, gws, kGemmScratch, stream));
    if (conv_implicit_ok(n, Cout, Cout, 3, 1, 9 * Cout, B, H, W)) {      // as fpn.py does for the same shapes (same bits, no column matrix)
        const int rc = conv3x3_implicit(A, y, hd.n1_w, hd.n1_b, 1e-6f, hd.w3, nullptr, y2, H, W, B, 1, Cout, Cout, stream);
        if (rc != FO1_OK) return rc;
        FO1_RUN(fo1_layernorm_bf16(y2, Cout, hd.n3_w, hd.n3_b, dst, Cout, n, Cout, 1e-6f, stream));
    } else {
        void* col = A.take(bf16_rows(n, 9 * Cout));
        FO1_RUN(fo1_layernorm_bf16(y, Cout, hd.n1_w, hd.n1_b, y2, Cout, n, Cout, 1e-6f, stream));
        FO1_RUN(fo1_im2col_bf16(y2, col, H, W, Cout, 3, 3, 1, 1, 9 * Cout, B, stream));
        FO1_RUN(fo1_gemm_bf16_ws(col, 9 * Cout, hd.w3, 9 * Cout, nullptr, nullptr, 0, y, Cout, n, Cout, 9 * Cout, 0, 0, gws, kGemmScratch, stream));
        FO1_RUN(fo1_layernorm_bf16(y, Cout, hd.n3_w, hd.n3_b, dst, Cout, n, Cout, 1e-6f, stream));
    }
    A.release(m);
    return FO1_OK;
}

// ConvTranspose2d(k=2, s=2) = GEMM [n, Cin] x [4 Cout, Cin]^T + pixel shuffle      (simple_fpn.py:141-150)
int fpn_up(Arena& A, const void* x, int Cin, const void* w, const void* b, int Cout, void* dst, int H, int W, int B, void* gws, void* stream) {
    const size_t m = A.mark();
    const int n = B * H * W;
    void* g = A.take(bf16_rows(n, 4 * Cout));
    FO1_RUN(fo1_gemm_bf16_ws(x, Cin, w, Cin, b, nullptr, 0, g, 4 * Cout, n, 4 * Cout, Cin, 0, 0, gws, kGemmScratch, stream));
    FO1_RUN(fo1_pixel_shuffle2_bf16(g, dst, H, W, Cout, B, stream));
    A.release(m);
    return FO1_OK;
}

int fpn_run(Arena& A, const fo1_fpn_weights_t* w, const void* x, int H, int W, int B, void* const* outs, void* stream) {
    const int Cin = w->c_in, c1 = w->c_up1, c2 = w->c_up2, Co = w->c_out, n = B * H * W;
    void* gws = A.take(kGemmScratch);
    int rc;
    {   // level 0: two 2x up-convolutions (LN + GELU between) -> head at (4H, 4W)
        const size_t m = A.mark();
        void* u1 = A.take(bf16_rows(4LL * n, c1));
        void* u1n = A.take(bf16_rows(4LL * n, c1));
        void* u2 = A.take(bf16_rows(16LL * n, c2));
        if ((rc = fpn_up(A, x, Cin, w->t1a_w, w->t1a_b, c1, u1, H, W, B, gws, stream)) != FO1_OK) return rc;
        FO1_RUN(fo1_layernorm_bf16(u1, c1, w->t1_ln_w, w->t1_ln_b, u1n, c1, 4 * n, c1, 1e-6f, stream));
        FO1_RUN(fo1_bias_act_bf16(u1n, c1, nullptr, u1, c1, 4 * n, c1, 1, stream));
        if ((rc = fpn_up(A, u1, c1, w->t1b_w, w->t1b_b, c2, u2, 2 * H, 2 * W, B, gws, stream)) != FO1_OK) return rc;
        if ((rc = fpn_head(A, w->heads[0], u2, c2, outs[0], 4 * H, 4 * W, B, Co, gws, stream)) != FO1_OK) return rc;
        A.release(m);
    }
    {   // level 1: one up-convolution -> head at (2H, 2W)
        const size_t m = A.mark();
        void* u = A.take(bf16_rows(4LL * n, c1));
        if ((rc = fpn_up(A, x, Cin, w->t2_w, w->t2_b, c1, u, H, W, B, gws, stream)) != FO1_OK) return rc;
        if ((rc = fpn_head(A, w->heads[1], u, c1, outs[1], 2 * H, 2 * W, B, Co, gws, stream)) != FO1_OK) return rc;
        A.release(m);
    }
    if ((rc = fpn_head(A, w->heads[2], x, Cin, outs[2], H, W, B, Co, gws, stream)) != FO1_OK) return rc;
    {   // level 3: 2x2 max-pool -> head at (H/2, W/2)
        const size_t m = A.mark();
        void* p = A.take(bf16_rows((long long)B * (H / 2) * (W / 2), Cin));
        FO1_RUN(fo1_maxpool2_bf16(x, p, H, W, Cin, B, stream));
        if ((rc = fpn_head(A, w->heads[3], p, Cin, outs[3], H / 2, W / 2, B, Co, gws, stream)) != FO1_OK) return rc;
        A.release(m);
    }
    return FO1_OK;
}
}  // namespace

size_t fo1_davit_workspace_bytes(const fo1_davit_weights_t* w, const fo1_davit_plan_t* plan) {
    if (!w || !plan) return 0;
    Arena A(nullptr);
    void* outs[4] = {(void*)256, (void*)256, (void*)256, (void*)256};
    davit_run(A, w, plan, nullptr, 0, outs, nullptr);
    return A.peak;
}

int fo1_davit_forward(const fo1_davit_weights_t* w, const fo1_davit_plan_t* plan, const void* images, int images_are_f32, void* const* maps_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    FO1_CHECK_ARG(w && plan && images && maps_out && workspace, "davit_forward: NULL argument");
    FO1_CHECK_ARG(w->n_stages >= 1 && w->n_stages <= 4 && plan->batch >= 1 && plan->H > 0 && plan->W > 0, "davit_forward: bad geometry");
    for (int i = 0; i < w->n_stages; ++i) FO1_CHECK_ARG(maps_out[i] && plan->items[i] && w->stages[i].blocks, "davit_forward: stage %d incomplete", i);
    const size_t need = fo1_davit_workspace_bytes(w, plan);
    if (workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "davit_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    Arena A(workspace);
    return davit_run(A, w, plan, images, images_are_f32, maps_out, stream);
}

size_t fo1_simplefpn_workspace_bytes(const fo1_fpn_weights_t* w, int H, int W, int batch) {
    if (!w || H <= 0 || W <= 0 || batch <= 0) return 0;
    Arena A(nullptr);
    void* outs[4] = {(void*)256, (void*)256, (void*)256, (void*)256};
    fpn_run(A, w, nullptr, H, W, batch, outs, nullptr);
    return A.peak;
}

int fo1_simplefpn_forward(const fo1_fpn_weights_t* w, const void* vit_map, int H, int W, int batch, void* const* maps_out, void* workspace,
                          size_t workspace_bytes, void* stream) {
    FO1_CHECK_ARG(w && vit_map && maps_out && workspace && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && batch >= 1, "simplefpn_forward: bad argument");
    for (int i = 0; i < 4; ++i) FO1_CHECK_ARG(maps_out[i] != nullptr, "simplefpn_forward: maps_out[%d] is NULL", i);
    const size_t need = fo1_simplefpn_workspace_bytes(w, H, W, batch);
    if (workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "simplefpn_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    Arena A(workspace);
    return fpn_run(A, w, vit_map, H, W, batch, maps_out, stream);
}

// mlpN_gelu projector (multimodal_projector/builder.py:64-71,103-110): Linear (+ GELU between layers), rows [M, dims[0]] -> [M, dims[n]]
size_t fo1_projector_workspace_bytes(const fo1_projector_t* p, int rows) {
    if (!p || rows <= 0 || p->n_layers < 1 || p->n_layers > 4) return 0;
    int wide = 0;
    for (int i = 1; i <= p->n_layers; ++i) wide = p->dims[i] > wide ? p->dims[i] : wide;
    return 2 * (((size_t)rows * wide * 2 + 255) & ~(size_t)255) + kGemmScratch;
}

int fo1_projector_forward(const fo1_projector_t* p, const void* x, int ldx, int rows, void* out, int ld_out, void* workspace, size_t workspace_bytes,
                          void* stream) {
    FO1_CHECK_ARG(p && x && out && workspace && rows > 0 && p->n_layers >= 1 && p->n_layers <= 4, "projector_forward: bad argument");
    const size_t need = fo1_projector_workspace_bytes(p, rows);
    if (workspace_bytes < need) return fo1::set_err(FO1_ERR_WORKSPACE, "projector_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    int wide = 0;
    for (int i = 1; i <= p->n_layers; ++i) wide = p->dims[i] > wide ? p->dims[i] : wide;
    const size_t slab = ((size_t)rows * wide * 2 + 255) & ~(size_t)255;
    char* base = (char*)workspace;
    void* gws = base + 2 * slab;
    const void* cur = x;
    int ld = ldx;
    for (int i = 0; i < p->n_layers; ++i) {
        const bool last = (i + 1 == p->n_layers);
        void* dst = last ? out : (void*)(base + (i & 1) * slab);
        const int ldd = last ? ld_out : p->dims[i + 1];
        FO1_TRY(fo1_gemm_bf16_ws(cur, ld, p->w[i], p->dims[i], p->b[i], nullptr, 0, dst, ldd, rows, p->dims[i + 1], p->dims[i], last ? 0 : 1, 0, gws, kGemmScratch,
                                 stream));
        cur = dst;
        ld = ldd;
    }
    return FO1_OK;
}

}  // extern "C"
