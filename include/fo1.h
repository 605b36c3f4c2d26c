/*
 * fo1.h — C-ABI of libfo1hip.so, the MI355X (gfx950) engine for the VLM-FO1 hot path.
 *
 * Plain C: raw device pointers, explicit shapes/strides, an opaque stream handle.
 * No torch / HIP types in any signature.  Every entry point returns int:
 *   0 = ok, <0 = argument/shape error (text via fo1_last_error()), >0 = hipError_t.
 * The library never allocates or frees device memory: outputs and workspaces are
 * caller-owned (PyTorch's caching allocator in the Python host).  Inputs are const.
 * All kernels are enqueued asynchronously on `stream` (a hipStream_t cast to void*;
 * NULL = the legacy default stream).
 *
 * Each function cites the reference interface it replaces (paths relative to
 * om-ai-lab/VLM-FO1 @ 2025-10-31).  INTEGRATION.md shows the ctypes binding the
 * reference-side maintainer would add.
 */
#ifndef FO1_H
#define FO1_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default) /* libfo1hip*.so are built with -fvisibility=hidden: exactly the declarations of this header are exported */
#endif

#define FO1_ABI_VERSION 9   /* 9: fo1_window_attention_bf16 (DaViT window attention on the q/k/v rows: no V^T copy), fo1_window_attention_map_bf16 / _var (the same on un-partitioned pixel rows: no window partition / reverse), fo1_attention_windows_bf16 (single-tile work lists in a software pipeline; fo1_vit_plan_t.q_block_win 0 selects it); 8: fo1_attention_decode_batch_partials_bf16 + fo1_gemv_attn_combine_bf16 (decode step at <= 2 sequences: the o-projection sums the split-KV partials in its prologue, no combine launch); 7: fo1_vit_block_t gained wqkv_hm / bqkv_hm (optional head-major q/k/v copy: fo1_vit_forward then takes the fused q/k/v epilogue); 6: attention q_block 128 / 256 (32x32-MFMA prefill kernel); fo1_qkv_proj_rope_bf16 (q/k/v projection with RoPE / K append / V^T in the GEMM epilogue); fo1_gemm_bf16_wtiled, fo1_splitk_swiglu_bf16 (measured no-gain forms), fo1_mfma_clock_probe, fo1_gemm_profile_shapes (instruments) moved to fo1_ab.h; 5: split-K planes consumed by fused kernels in the decode pool (fo1_gemm_bf16_partials, fo1_splitk_residual_rmsnorm_bf16, fo1_pool_qkv_post_partials_bf16, fo1_splitk_swiglu_bf16), fo1_gemm_bf16_wtiled, fo1_mfma_clock_probe; 4: decode pool (fo1_pool_qkv_post_bf16; fo1_decode_argmax_accept up to 256 rows): continuous batching of 33..128 sequences; 3: fo1_hfre_opts_t grew out_bf16 / out_bf16_ld; fo1_img_seg + the *_var spatial entry points */
#define FO1_OK 0
#define FO1_ERR_ARG (-1)       /* bad argument / unsupported shape */
#define FO1_ERR_WORKSPACE (-2) /* workspace too small */

int fo1_abi_version(void);
/* Thread-local text of the last error returned on this thread ("" if none). */
const char* fo1_last_error(void);

/* ------------------------------------------------------------------------
 * Per-kernel timing for bench.py's roofline line.  While enabled, every kernel the
 * library launches goes through hipExtLaunchKernelGGL with a start/stop event pair on its
 * launch stream (kernel execution time only, the clocks rocprofv3's kernel trace reads).
 * Off by default, during throughput timing and during graph capture.
 * fo1_profile_stage(tag) labels the records since the previous call "<tag>|<kernel>" (stage
 * boundaries for the per-stage breakdown); it does not synchronise.
 * `total_work` accumulates the ALGORITHMIC bytes (HBM-bound kernels) or flops
 * (MFMA-bound kernels) of each launch, as documented per kernel in DESIGN.md.
 * ---------------------------------------------------------------------- */
typedef struct fo1_profile_row {
    char name[48];
    int64_t calls;
    double total_ms;
    double total_work;
} fo1_profile_row_t;
int fo1_profile_enable(int on);
int fo1_profile_read(fo1_profile_row_t* rows, int cap, int reset);
int fo1_profile_stage(const char* tag);

/* ------------------------------------------------------------------------
 * HFRE region pooling  
 *
 * Replaces the arithmetic of
 *   HFREModule.__call__                 hybrid_finegrained_region_encoder.py:275-469
 *   HFREModule.extract_vt_region_feature                                   :230-273
 *   gen_sineembed_for_position                                             :55-103
 *   the three torchvision.ops.roi_align call sites                         :248,263,353
 *   F.interpolate(bilinear) + torch.cat of the aux pyramid                 :338-350
 *   the aux→vt box scaling in encode_regions      omchat_qwen2_5_vl.py:94-99
 * for the product configuration (use_vision_tower_region_feature, 'concat',
 * 'bbox_based').  One call = all boxes of one image, all feature sources.
 *
 * out[n, src.out_offset + c] =
 *     mean_{7x7 bins}( roi_align(src map (bilinearly upsampled to roi_H x roi_W when
 *                       they differ), box_n * src.spatial_scale, aligned=False,
 *                       sampling_ratio=-1) )[c]
 *   + sine box embedding (pos_mode != 0)
 * computed with the exact separable form (per-axis tap weights composed with the
 * per-axis upsample matrix), so each source map is read once at native
 * resolution in bf16 and no [C,H0,W0] fp32 intermediate exists.
 * ---------------------------------------------------------------------- */
#define FO1_HFRE_MAX_SOURCES 8
#define FO1_HFRE_MAX_EXTENT 1024 /* max H or W of any source / roi map */

typedef struct fo1_hfre_source {
    const void* data;    /* device, bf16, token-major (channels-last): element (h,w,c) at
                            data[(h*W + w)*ld + c]                                        */
    int32_t H, W;        /* map size                                                       */
    int32_t C;           /* channels pooled from this source (multiple of 64)              */
    int32_t ld;          /* row stride in elements (>= C, multiple of 8; data 16-B aligned)*/
    int32_t roi_H, roi_W;/* size of the map roi_align runs on; != (H,W) means the source is
                            bilinearly upsampled (align_corners=False) to it first         */
    float spatial_scale; /* roi_align spatial_scale on the roi_H x roi_W map               */
    int32_t box_space;   /* 0: aux boxes as given; 1: vt boxes = aux * (vt_scale_x, _y)    */
    int32_t out_offset;  /* first output channel written by this source                    */
} fo1_hfre_source_t;

/* Bytes of scratch fo1_hfre_region_pool needs for these sources / n_boxes. */
size_t fo1_hfre_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes);

int fo1_hfre_region_pool(
    const fo1_hfre_source_t* sources, int n_sources, /* host array, copied at launch */
    const float* boxes_aux, int n_boxes,             /* device fp32 [n_boxes,4] xyxy, aux px */
    const float* boxes_vt,                           /* device fp32 [n_boxes,4] in vt px, or NULL:
                                                        then vt box = aux box * (vt_scale_x,_y),
                                                        one fp32 multiply as omchat_qwen2_5_vl.py:99 */
    float vt_scale_x, float vt_scale_y,
    int roi_size,                                    /* pooled bins per axis (7)             */
    int pos_mode,                                    /* 0 none; 1 embed vt boxes; 2 aux boxes*/
    float pos_img_w, float pos_img_h,                /* box normalisers (grid*14 / aux size) */
    float* out, int out_ld, int region_dim,          /* device fp32 [n_boxes, out_ld]        */
    void* workspace, size_t workspace_bytes,         /* device scratch                       */
    void* stream);

/* Work-list HFRE for the boxes of `batch` same-geometry images at once: the tap-weight kernel also lists the (box, source, chunk,
 * slice) tuples that exist, a fixed grid walks that list (no empty workgroups — nine in ten of fo1_hfre_region_pool's are), and the
 * row finish applies the optional region LayerNorm.  Same arithmetic as fo1_hfre_region_pool; a box's result does not depend on
 * the other boxes or images of the call.  opts may be NULL (one image, no LayerNorm).  Region LayerNorm =
 * mm_apply_region_layer_norm (reference hybrid_finegrained_region_encoder.py:365-372): fp32 nn.LayerNorm of the aux block
 * [0, ln_split) and the vt block [ln_split, region_dim) before the position embedding.
 * Workspace contract: its first 64 bytes (the work-list counter) must be zero the first time a workspace is used; every call leaves
 * them zero (the last kernel resets the counter in-stream — no memset node, so the call is safe to capture in a hipGraph). */
typedef struct fo1_hfre_opts {
    int32_t batch;                               /* images in this call (>= 1)                                          */
    const int32_t* box_image;                    /* device int32[n_boxes]: image of each box (NULL when batch == 1)     */
    long long img_stride[FO1_HFRE_MAX_SOURCES];  /* elements from image b to image b+1 of source i (multiple of 8)      */
    int32_t ln_on, ln_split;
    const float* ln_w0; const float* ln_b0;      /* fp32 device, block [0, ln_split)                                    */
    const float* ln_w1; const float* ln_b1;      /* fp32 device, block [ln_split, region_dim)                           */
    float ln_eps;
    void* out_bf16; int32_t out_bf16_ld;         /* optional: the rows ALSO written as bf16 (round to nearest even) — the cast
                                                    encode_regions applies before mm_projector_aux (omchat_qwen2_5_vl.py:106);
                                                    device bf16 [n_boxes, ld >= region_dim], ld % 4 == 0; NULL = fp32 only      */
} fo1_hfre_opts_t;
size_t fo1_hfre_ex_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes);
int fo1_hfre_region_pool_ex(const fo1_hfre_source_t* sources, int n_sources, const float* boxes_aux, int n_boxes,
                               const float* boxes_vt, float vt_scale_x, float vt_scale_y, int roi_size, int pos_mode,
                               float pos_img_w, float pos_img_h, float* out, int out_ld, int region_dim,
                               const fo1_hfre_opts_t* opts, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * bf16 GEMM with fused epilogue  (SURVEY §8a rows a2,a4,a5,a8,a9,a11: every nn.Linear /
 * 1x1-conv / patch-embed of the hot path; reference call sites
 * modeling_qwen2_5_vl.py:79-81,103-110,151-155,176-177,633-635,731-734,
 * modeling_davit.py:63-65,157-158,235-236, multimodal_projector/builder.py:64-71,103-110)
 *
 *   C[M,N] = epilogue(A[M,K] . W[N,K]^T),  W = nn.Linear weight [out,in], bf16 in, fp32 accumulate
 *   epilogue (each step rounds to bf16 like the reference's op-by-op bf16 tensors):
 *     y = bf16(acc + bias[n]);  y = bf16(act(y));  y = bf16(y + residual[m,n])
 *   act: 0 none, 1 exact-erf GELU, 2 SiLU, 3 fused SwiGLU: W rows (and bias) interleaved in 16-row groups
 *        [gate 16 | up 16 | ...] (N = 2F, N % 32 == 0); C[M, F] = bf16(bf16(silu(bf16(gate))) * bf16(up)) —
 *        act_fn(gate_proj(x)) * up_proj(x) of modeling_qwen2_5_vl.py:85-86,636 without the [M,2F] round trip.
 *   out_f32 != 0: C is fp32 (acc + bias, act), no residual.
 * K, lda, ldw multiples of 8; A, W 16-byte aligned.  bias/residual may be NULL.
 * ---------------------------------------------------------------------- */
int fo1_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias,
                  const void* residual, int ldr, void* C, int ldc,
                  int M, int N, int K, int act, int out_f32, void* stream);
/* Same, with a caller-owned fp32 scratch (16-byte aligned) that enables split-K for skinny outputs:
 * partials [splits][M][N] are reduced in a fixed order by a second kernel (deterministic). */
int fo1_gemm_bf16_ws(const void* A, int lda, const void* W, int ldw, const void* bias,
                     const void* residual, int ldr, void* C, int ldc,
                     int M, int N, int K, int act, int out_f32,
                     void* workspace, size_t workspace_bytes, void* stream);
/* ----------------------------------------------------------------------
 * fp8 linear — BASELINE configs[4] ("fp8 MFMA").  The reference has no fp8 path (it runs bf16 everywhere, builder.py:40-46):
 * this is the W8A8 form of the same nn.Linear call sites, per-token activation scales x per-output-channel weight scales.
 *   fo1_quantize_rows_e4m3: q[m, :] = e4m3fn(clamp(x[m, :] / scales[m], +-448)), scales[m] = absmax(x[m, :]) / 448 (1 for a zero
 *     row); OCP e4m3fn, round to nearest even.  x bf16 [M, K] (ldx elements), q bytes [M, K] (ldq bytes), K % 8 == 0.
 *   fo1_gemm_fp8: C[M, N] = epilogue((Aq Wq^T) * scale_a[m] * scale_w[n]): v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
 *     scales, fp32 accumulation, the bf16 epilogues of fo1_gemm_bf16 (act 0 none, 1 GELU, 2 SiLU, 3 interleaved SwiGLU; bias,
 *     residual).  K % 128 == 0, lda / ldw % 16 == 0 (bytes), operands 16-byte aligned, each operand < 4 GB.
 * ---------------------------------------------------------------------- */
int fo1_quantize_rows_e4m3(const void* x, long long ldx, int M, int K, void* q, long long ldq, float* scales, void* stream);
/* Qwen2RMSNorm (modeling_qwen2_5_vl.py:126-140) + the quantiser above in one launch: bit-identical to fo1_rmsnorm_bf16 followed by
 * fo1_quantize_rows_e4m3, the bf16 row never reaches memory. */
int fo1_rmsnorm_quant_e4m3(const void* x, int ldx, const void* weight, int M, int D, float eps, void* q, long long ldq, float* scales,
                           void* stream);
int fo1_gemm_fp8(const void* Aq, int lda, const float* scale_a, const void* Wq, int ldw, const float* scale_w, const void* bias,
                 const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* stream);

/* ------------------------------------------------------------------------
 * Row norms / small elementwise ops (HBM-bound).  All tensors bf16 row-major with explicit row
 * strides (elements, multiples of 8); D multiple of 8, <= 4096.
 *   fo1_rmsnorm_bf16   Qwen2RMSNorm              modeling_qwen2_5_vl.py:126-140 (ViT blocks, merger, LLM)
 *   fo1_layernorm_bf16 nn.LayerNorm              modeling_davit.py:29-48 (PreNorm), :135-141 (ConvEmbed)
 *                      and the channel LayerNorm of simple_fpn.py:58-78 on token-major maps
 *   fo1_swiglu_bf16    act_fn(gate)*up           modeling_qwen2_5_vl.py:85-86, :636; input rows are [gate | up]
 *   fo1_bias_act_bf16  y = act(x + bias)         act 0 none / 1 erf-GELU (nn.GELU, simple_fpn.py:145)
 *   fo1_argmax_bf16    greedy next token         first index among ties, like torch.argmax (two-stage over 128
 *                                                slices when a 1 KiB scratch is given)
 * ---------------------------------------------------------------------- */
int fo1_rmsnorm_bf16(const void* x, int ldx, const void* weight, void* y, int ldy, int M, int D,
                     float eps, void* stream);
int fo1_layernorm_bf16(const void* x, int ldx, const void* weight, const void* bias, void* y, int ldy,
                       int M, int D, float eps, void* stream);
/* fo1_layernorm_bf16 with output row m written to row y_rows[m] of y (int32 [M]): the producer side of fo1_conv3x3_gemm_bf16's padded map. */
int fo1_layernorm_rows_bf16(const void* x, int ldx, const void* weight, const void* bias, void* y, int ldy, const int32_t* y_rows, int M, int D,
                            float eps, void* stream);
int fo1_swiglu_bf16(const void* gate_up, int ldgu, void* out, int ldo, int M, int F, void* stream);
int fo1_bias_act_bf16(const void* x, int ldx, const void* bias, void* y, int ldy, int M, int D, int act,
                      void* stream);
int fo1_argmax_bf16(const void* x, int n, int* out, void* scratch /* 1 KiB device scratch or NULL */, void* stream);

/* ------------------------------------------------------------------------
 * Rotary embeddings on the fused QKV activations, and the V -> V^T copy.
 *   fo1_rope_llm_bf16  apply_multimodal_rotary_pos_emb  modeling_qwen2_5_vl.py:643-685 with the
 *        section-selected bf16 cos/sin [L, head_dim] of :603-624; rotates heads [0,n_heads) of width
 *        head_dim starting at column col0 in place; heads >= k_first_head are also appended to
 *        kcache[(head-k_first_head)*kcache_head_stride + (pos0+t)*head_dim] when kcache != NULL.
 *   fo1_rope_vit_bf16  apply_rotary_pos_emb_flashatt / _vision  :162-169, :219-230: fp32 cos/sin
 *        [S, head_dim/2]; rotates the q and k heads (2*n_heads heads from column 0) in place.
 *   fo1_transpose_bf16 dst[c*ld_dst + col0 + m] = src[m*ld_src + c]   (C multiple of 64; col0 from *dyn_col0 if set)
 *   fo1_qkv_post_llm_bf16 / fo1_qkv_post_vit_bf16  prefill: the rope above on the q/k heads of the fused qkv rows
 *        [q heads | k heads | v heads] PLUS the V -> V^T copy (and, LLM, the K-cache append at pos0) in ONE launch.
 * ---------------------------------------------------------------------- */
int fo1_rope_llm_bf16(void* qkv, int ld, int col0, int n_heads, int head_dim, const void* cos_bf16,
                      const void* sin_bf16, int L, void* kcache, int k_first_head,
                      long long kcache_head_stride, int pos0, const int32_t* dyn_state, void* stream);
int fo1_rope_vit_bf16(void* qkv, int ld, int n_heads, int head_dim, const float* cos_f32,
                      const float* sin_f32, int S, void* stream);
int fo1_transpose_bf16(const void* src, int ld_src, void* dst, long long ld_dst, int col0,
                       const int32_t* dyn_col0, int M, int C, void* stream);
int fo1_qkv_post_llm_bf16(void* qkv, int ld, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_bf16,
                          const void* sin_bf16, int L, void* kcache, long long kcache_head_stride,
                          void* vtcache, long long vt_row_stride, int pos0, void* stream);
int fo1_qkv_post_vit_bf16(void* qkv, int ld, int n_heads, int head_dim, const float* cos_f32,
                          const float* sin_f32, int S, void* vt, long long vt_ld, void* stream);
/* Decode-step bookkeeping kept on the device so that ONE captured hipGraph serves every generated token:
 * state = device int32[8]: [0] cache position, [1] rope-table row (position + rope delta), [4..7] the attention
 * work item {pos, pos+1, 0, pos+1}.  fo1_rope_llm_bf16 (dyn_state), fo1_transpose_bf16 (dyn_col0 = &state[0]) and
 * fo1_attention_bf16 (items = &state[4], q_row_base = &state[0]) read it; fo1_decode_advance increments it.
 * (reference: cache_position + rope_deltas arithmetic of modeling_qwen2_5_vl.py:1848-1860) */
int fo1_decode_advance(int32_t* state, void* stream);
/* One kernel for the decode step's q/k/v post-processing: mRoPE on the q and k heads of the fused qkv row (in place,
 * table row = state[1]), K heads appended to the K cache and V heads (transposed) to the V^T cache at position state[0]. */
int fo1_decode_qkv_post_bf16(void* qkv_row, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                             const void* sin_table, const int32_t* state, void* kcache, long long kcache_head_stride,
                             void* vtcache, long long vt_row_stride, void* stream);
/* The same for the P sequences of a decode pool: row b of qkv [P, ld] is rotated with table row state[b][1] (state = int32[P][8],
 * the batched-decode layout below); its K heads / V heads land at cache row / V^T column state[b][0] (the sequence's own slot). */
int fo1_pool_qkv_post_bf16(void* qkv, long long ld, int P, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                           const void* sin_table, const int32_t* state, void* kcache, long long kcache_head_stride,
                           void* vtcache, long long vt_row_stride, void* stream);
/* The same fed by the split-K planes of the q/k/v projection (fo1_gemm_bf16_partials below): row b = bf16(sum_z part[z][b] + bias), the
 * GEMM epilogue's rounding; the rotated q heads go to q_out [P, ld] (the decode attention's query rows), K / V^T straight to the caches. */
int fo1_pool_qkv_post_partials_bf16(const float* part, int splits, const void* bias, void* q_out, long long ld, int P, int n_q_heads,
                                    int n_kv_heads, int head_dim, const void* cos_table, const void* sin_table, const int32_t* state,
                                    void* kcache, long long kcache_head_stride, void* vtcache, long long vt_row_stride, void* stream);
/* Decode-pool projections as split-K planes (Qwen2_5_VLDecoderLayer's q/k/v, o and down nn.Linear at 33..128 rows, modeling_qwen2_5_vl.py
 * :636, :700-742): part[z][m][n] (fp32, row stride N) = A[m, K run z] . W[n, K run z], z < *splits_out (the effective count for the requested
 * `splits`: K tiles of 64 in equal runs).  No epilogue, no reduce launch — a few-row product has too few output tiles to fill 256 CUs, and
 * its consumer reads the planes anyway:
 *   fo1_splitk_residual_rmsnorm_bf16   x_out = bf16(bf16(sum_z part[z] (+ bias)) + residual);  xn_out = Qwen2RMSNorm(x_out) * norm_weight
 *                                      (the residual add of :736 / :742 and the NEXT layernorm, :728 / :739, in the launch that reduces)
 *   fo1_pool_qkv_post_partials_bf16    above.
 * K % 64 == 0, N % 4 == 0 (% 8 for the consumers), operands 16-byte aligned, part holds splits * M * N floats. */
int fo1_gemm_bf16_partials(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int splits, float* part, int* splits_out,
                           void* stream);
int fo1_splitk_residual_rmsnorm_bf16(const float* part, int splits, int M, int N, const void* bias, const void* residual, int ldr, void* x_out,
                                     int ldx, const void* norm_weight, float eps, void* xn_out, int ldn, void* stream);
/* q/k/v projection + bias + rotary embedding + K-cache append + V^T write in ONE launch (round 5): exactly what fo1_gemm_bf16 followed by
 * fo1_qkv_post_llm_bf16 / fo1_qkv_post_vit_bf16 computes, bit for bit, without the second pass over the [M, N] activation — the rotation, the cache
 * append and the transposed V store run in the 256 x 256 GEMM kernel's epilogue while the tile is still in LDS.
 *   mode 0, LLM (modeling_qwen2_5_vl.py:643-685,731-734): W rows [q heads | k heads | v heads] of head_dim 128; cos / sin bf16 [M][128] (the packed rows'
 *     mRoPE tables); rotated q -> C[:, :n_q_heads * 128] (the k / v columns of C are not written); rotated k -> kcache[kv head][pos0 + m][128];
 *     v -> vt[(kv head * 128 + d) * vt_ld + pos0 + m].
 *   mode 1, ViT (:162-169,219-230): W rows HEAD-MAJOR — per head [q 80 | k 80 | v 80 | 16 zero rows], N = 256 * n_q_heads (a head's rotate-half
 *     pairs then never straddle two output tiles); cos / sin fp32 [M][40]; rotated q, k -> C in that layout (attention: head stride 256, k at
 *     column 80 of a head); v -> vt[(head * 80 + d) * vt_ld + pos0 + m].  kcache unused.
 * K % 64 == 0, N % 256 == 0, pos0 % 8 == 0, vt_ld % 8 == 0, 16-byte aligned operands; any M (built for M >= 1024: one 256 x 256 tile per workgroup). */
/* 3x3 convolution as an IMPLICIT GEMM (round 5; DaViT ConvEmbed modeling_davit.py:102-148, SimpleFPN simple_fpn.py:141-176): what fo1_im2col_bf16 +
 * fo1_gemm_bf16 compute — same 256 x 256 kernel, same K order (ky, kx, channel): bit-identical — without the [M, 9 Cin] column matrix.
 *   Xpad    token-major bf16 map(s), zero-padded by one pixel per side, row pitch Wp pixels of Cin channels (fo1_layernorm_rows_bf16 writes the
 *           normalised map straight into this layout; the caller zeroes the buffer once)
 *   a_rows  uint32 [M]: byte offset in Xpad of output pixel m's top-left tap — built by the host for any stride / batch of images sharing Wp
 *   W       [N][3][3][Cin] bf16 (row stride ldw >= 9 Cin), Cin = 64 * 2^j;  act 0 / 1 (GELU);  C [M, N] bf16. */
int fo1_conv3x3_gemm_bf16(const void* Xpad, const uint32_t* a_rows, int Wp, int Cin, const void* W, int ldw, const void* bias, void* C, int ldc, int M,
                          int N, int act, void* stream);
/* 1 when fo1_gemm_bf16 runs an [M, K] x [N, K]^T bf16 product (K % 64 == 0, aligned operands) on the 256 x 256 kernel: where a caller may swap
 * fo1_gemm_bf16 + fo1_qkv_post_* for fo1_qkv_proj_rope_bf16 without changing a bit. */
int fo1_gemm_takes_big_tile(int M, int N, int K);
int fo1_qkv_proj_rope_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K, int mode,
                           int n_q_heads, int n_kv_heads, const void* cos_table, const void* sin_table, void* kcache, long long kcache_head_stride,
                           int pos0, void* vt, long long vt_ld, void* stream);
/* Weight-streaming GEMV (M <= 4) with the fo1_gemm_bf16 epilogues and an optional fused Qwen2RMSNorm on the input rows
 * (norm_weight [K] or NULL): folds input_layernorm / post_attention_layernorm into the projections of the decode step. */
int fo1_gemv_bf16(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                  void* C, int ldc, int M, int N, int K, int act, const void* norm_weight, float norm_eps, void* stream);

/* ------------------------------------------------------------------------
 * Fused attention  softmax(scale * Q K^T [+ causal mask]) V   (flash_attn_varlen_func /
 * _flash_attention_forward / SDPA at modeling_qwen2_5_vl.py:205,319,895,990; DaViT window
 * attention modeling_davit.py:262-270).  head_dim in {32, 80, 128}; GQA via n_q_heads/n_kv_heads.
 * `items` = device int32[n_items][4] {q_start, q_end, kv_start, kv_end}: each item is <= q_block
 * queries attending keys [kv_start, kv_end) (and key <= query when causal); kv_start % 4 == 0.
 * q_block 16, 32 or 64: 16x16-MFMA kernel, one wave per 16 queries (smaller blocks = more workgroups
 * for short sequences).  q_block 128 or 256 (head_dim 80 / 128, no q_row_base): 32x32-MFMA kernel,
 * 8 waves x 32 queries per workgroup — 256 queries of one head, or 128 queries x the TWO query heads
 * of one KV head (n_q_heads / n_kv_heads even: both heads share the staged K / V^T tiles); O rows are
 * stored in 16-byte pieces (O 16-byte aligned, strides % 8 == 0), K / V^T row strides < 2^22, and the
 * key rows are addressed with unsigned 32-bit byte offsets from the head's K base: every item's
 * kv_end * k_tok_stride * 2 must stay below 2^32 (the items live on the device, so the CALLER checks:
 * vlm_fo1_amd/ops.py pick_q_block / _check_attn32_extent fall back to q_block 64 past that).
 * V is passed transposed: VT[(kv_head*head_dim + d)*vt_row_stride + key] (fo1_transpose_bf16),
 * finite beyond kv_end up to the next multiple of 4.  Strides in elements.
 * ---------------------------------------------------------------------- */
int fo1_attention_bf16(const void* Q, long long q_tok_stride, long long q_head_stride,
                       const void* K, long long k_tok_stride, long long k_head_stride,
                       const void* VT, long long vt_row_stride,
                       void* O, long long o_tok_stride, long long o_head_stride,
                       const int32_t* items, int n_items, int q_block, int n_q_heads, int n_kv_heads,
                       int head_dim, float scale, int causal, const int32_t* q_row_base, double flops_hint,
                       void* stream);
/* fo1_attention_bf16 for work lists of SINGLE-TILE items (round 6; the ViT's 112-pixel windows, modeling_qwen2_5_vl.py:172-209): every item has at
 * most 64 keys, its queries inside its key range, no causal mask.  A workgroup walks 4 consecutive items of the list for its head and requests
 * item i + 1's K / V^T tile and queries while item i is computed (fo1_attention_bf16 runs such an item as one dependent chain with nothing to
 * overlap).  Bit-identical to fo1_attention_bf16 (q_block 64) on the same list.  head_dim 80; o_rows = rows of O (32-bit store offsets: the
 * output at most 2 GiB). */
int fo1_attention_windows_bf16(const void* Q, long long q_tok_stride, long long q_head_stride, const void* K, long long k_tok_stride,
                                long long k_head_stride, const void* VT, long long vt_row_stride, void* O, long long o_tok_stride, long long o_head_stride,
                                long long o_rows, const int32_t* items, int n_items, int n_q_heads, int n_kv_heads, int head_dim, float scale,
                                double flops_hint, void* stream);
/* fo1_attention_bf16 with a second key range per item: prefix_ranges int32 [n_items][2] = [start, end) (empty when start >= end), attended
 * in full by every query of the item before its own (causal) range.  Several prompts over ONE image share the rows of their common
 * prefix (system text + image tokens): the prefix runs through a layer once, each prompt's remaining rows attend [prefix | own rows].
 * The prefix rows precede the item's rows in the index space.  (Reference: one whole-model run per prompt of <= 100 regions,
 * mm_utils.py:600; BASELINE configs[4]'s 300 proposals are three prompts over one image.) */
int fo1_attention_prefix_bf16(const void* Q, long long q_tok_stride, long long q_head_stride,
                              const void* K, long long k_tok_stride, long long k_head_stride,
                              const void* VT, long long vt_row_stride,
                              void* O, long long o_tok_stride, long long o_head_stride,
                              const int32_t* items, const int32_t* prefix_ranges, int n_items, int q_block, int n_q_heads,
                              int n_kv_heads, int head_dim, float scale, int causal, double flops_hint, void* stream);

/* ------------------------------------------------------------------------
 * DaViT / SimpleFPN / splice data-movement kernels on token-major bf16 maps [H*W, C] (C % 8 == 0).
 * ABI 2: every spatial kernel takes `batch` — that many same-size images stacked along the row dimension
 * ([batch*H*W, C]; image b = rows [b*H*W, (b+1)*H*W)); neighbourhoods never cross an image boundary.  This is the
 * multi-image prefill of SURVEY 8f-3 (the reference's splice is batch-aware, omchat_qwen2_5_vl.py:380-416).
 *   fo1_dwconv3x3_bf16          y = x + bf16(dwconv3x3(x) + bias)   DepthWiseConv2d inside PreNorm(None,.)
 *                               modeling_davit.py:72-99,29-48; weight re-laid out [9][C] (tap-major)
 *   fo1_im2col_bf16             col[(oy,ox), (ky,kx,c)] for ConvEmbed (:102-148) and the FPN 3x3 conv
 *                               (simple_fpn.py:165-175); the conv itself is fo1_gemm_bf16 on col
 *   fo1_window_partition_bf16   zero-pad to multiples of ws and regroup rows window by window (:208-213,244-254)
 *   fo1_window_reverse_add_bf16 y = shortcut + window_reverse(yw)[:H,:W]   (:216-222,272-281 + PreNorm residual)
 *   fo1_window_attention_bf16   WindowAttention's softmax(q k^T * scale) v (:225-282) for head dim 32, straight on the q/k/v GEMM's
 *                               [windows * window_tokens, 3C] rows (q | k | v, heads 32 columns apart): one wave per (window, head),
 *                               no V^T copy, no item list (round 6; fo1_attention_bf16 + fo1_transpose_bf16 before)
 *   fo1_channel_attention_bf16  ChannelAttention (:151-172) for 32-wide groups: qkv rows [q|k|v] ->
 *                               out[n, g*32+c] = sum_c' softmax_c'(q_g^T k_g / sqrt(N))[c][c'] v[n, g*32+c']
 *   fo1_pixel_shuffle2_bf16     ConvTranspose2d(k=2,s=2) output regrouping (simple_fpn.py:141-150):
 *                               dst[(2y+dy, 2x+dx), co] = src[(y,x), (dy*2+dx)*Co + co]
 *   fo1_maxpool2_bf16           nn.MaxPool2d(2,2) (simple_fpn.py:153)
 *   fo1_nchw_to_hwc8_bf16       image [3,H,W] (bf16 or fp32) -> [H*W, 8] bf16, channels 3..7 zero
 *   fo1_gather_rows_bf16        out[r] = table[plan[r].kind][plan[r].index]  — embed_tokens + image/region
 *                               token splice (omchat_qwen2_5_vl.py:291-373); plan = int32[R][2]
 * ---------------------------------------------------------------------- */
int fo1_dwconv3x3_bf16(const void* x, const void* weight9c, const void* bias, void* y, int H, int W, int C,
                       int batch, void* stream);
/* fo1_dwconv3x3_bf16 followed by the LayerNorm every DaViT block applies to its result (modeling_davit.py:29-48 PreNorm after
 * :72-99 DepthWiseConv2d), one launch: y = x + dwconv(x) (the residual stream), h = LayerNorm(y).  Bit-identical to the two
 * separate calls.  C <= 2048. */
int fo1_dwconv3x3_ln_bf16(const void* x, const void* weight9c, const void* bias, void* y, const void* ln_weight,
                          const void* ln_bias, float ln_eps, void* h, int H, int W, int C, int batch, void* stream);
int fo1_im2col_bf16(const void* x, void* col, int H, int W, int C, int KH, int KW, int stride, int pad,
                    int ld_col, int batch, void* stream);
int fo1_window_partition_bf16(const void* x, void* xw, int H, int W, int C, int ws, int batch, void* stream);
int fo1_window_reverse_add_bf16(const void* yw, const void* shortcut, void* y, int H, int W, int C, int ws,
                                int batch, void* stream);
/* N = tokens of ONE image; qkv / out hold batch * N rows and every image gets its own per-group attention matrices.
 * Rows are read / written 16 bytes at a time: ld % 8 == 0, ldo % 8 == 0, qkv and out 16-byte aligned. */
/* out[w * window_tokens + i, h * 32 + d] for every window w < n_windows and head h < n_heads; C = n_heads * 32, window_tokens <= 160,
 * ld >= 3 C, rows 16-byte aligned (ld % 8, ldo % 8), the output at most 2 GiB (32-bit store offsets). */
int fo1_window_attention_bf16(const void* qkv, long long ld, int C, int n_heads, int window_tokens, int n_windows, void* out, long long ldo,
                              float scale, void* stream);
/* The same attention WITHOUT window partition / reverse (round 6): qkv holds the q/k/v rows of the images' PIXELS in raster order ([batch * H * W, ld],
 * the projection of the LayerNorm output itself), token (iy, ix) of window (wy, wx) is pixel (wy * window + iy, wx * window + ix); tokens outside the
 * image are the reference's zero padding after the norm (modeling_davit.py:248-251) and read `pad_row` = the projection of a zero row = the layer's
 * bf16 q/k/v bias [3C].  out [batch * H * W, ldo]: the pixels' rows (the proj GEMM then adds the residual in its own epilogue: no window_reverse).
 * window == 12, head dim 32.  _var: images of different sizes, segs = the fo1_img_seg table of fo1_window_partition_var_bf16 (in_row0, H, W, -, windows
 * down / across), max_windows = the most windows any image has. */
int fo1_window_attention_map_bf16(const void* qkv, long long ld, int C, int n_heads, int window, int H, int W, int batch, const void* pad_row, void* out,
                                  long long ldo, float scale, void* stream);
int fo1_window_attention_map_var_bf16(const void* qkv, long long ld, int C, int n_heads, int window, const void* segs, int n_img, int max_windows,
                                      long long total_pixels, const void* pad_row, void* out, long long ldo, float scale, void* stream);
size_t fo1_channel_attention_workspace_bytes(int N, int C, int batch);
int fo1_channel_attention_bf16(const void* qkv, int ld, int N, int C, void* out, int ldo, int batch, void* workspace,
                               size_t workspace_bytes, void* stream);
int fo1_pixel_shuffle2_bf16(const void* src, void* dst, int H, int W, int Co, int batch, void* stream);
int fo1_maxpool2_bf16(const void* x, void* y, int H, int W, int C, int batch, void* stream);
/* ---- ragged image batches (SURVEY 8f-3 on real datasets: CountBench / Pixmo / COCO images all differ in size) ----
 * The same kernels over images of DIFFERENT sizes packed row-wise into one map: `segs` is a DEVICE table of n_img records, one
 * workgroup column per image; every image is processed with exactly the arithmetic of the one-image call, so a ragged packed pass is
 * bit-identical to the one-image passes.  Record fields (per operator: `in_*` describes the operand, `out_*` the result):
 *   dwconv_ln / window_reverse / channel attention   in_row0 = first row of the image, H, W   (channel attention: H = its token count)
 *   im2col / maxpool                                 in_row0, H, W -> out_row0, Ho, Wo
 *   window_partition                                 in_row0, H, W -> out_row0 = first window row, Ho x Wo = windows down / across
 *   window_reverse_add                               out_row0 = first window row (yw), Ho x Wo = windows; in_row0 = pixels (shortcut, y)
 *   pixel_shuffle2                                   in_row0 (rows of [., 4*Co]), H, W -> out_row0 (rows of [., Co], 4*H*W of them)
 * max_* sizes the launch (largest image), total_* is the work figure of the profile rows.  The reference runs its towers image by
 * image (davit_aux_encoder.py:54-69, one forward per list element). */
typedef struct fo1_img_seg { int32_t in_row0, H, W, out_row0, Ho, Wo, aux0, aux1; } fo1_img_seg;
int fo1_dwconv3x3_ln_var_bf16(const void* x, const void* weight9c, const void* bias, void* y, const void* ln_weight, const void* ln_bias,
                              float ln_eps, void* h, const void* segs, int n_img, int max_pixels, long long total_pixels, int C, void* stream);
int fo1_im2col_var_bf16(const void* x, void* col, const void* segs, int n_img, int max_out_pixels, long long total_out_pixels, int C, int KH, int KW,
                        int stride, int pad, int ld_col, void* stream);
int fo1_window_partition_var_bf16(const void* x, void* xw, const void* segs, int n_img, int max_window_rows, long long total_window_rows, int C,
                                  int ws, void* stream);
int fo1_window_reverse_add_var_bf16(const void* yw, const void* shortcut, void* y, const void* segs, int n_img, int max_pixels, long long total_pixels,
                                    int C, int ws, void* stream);
size_t fo1_channel_attention_var_workspace_bytes(int max_tokens, int C, int n_img);
int fo1_channel_attention_var_bf16(const void* qkv, int ld, const void* segs, int n_img, int max_tokens, long long total_tokens, int C, void* out,
                                   int ldo, void* workspace, size_t workspace_bytes, void* stream);
int fo1_pixel_shuffle2_var_bf16(const void* src, void* dst, const void* segs, int n_img, int max_pixels, long long total_pixels, int Co, void* stream);
int fo1_maxpool2_var_bf16(const void* x, void* y, const void* segs, int n_img, int max_out_pixels, long long total_out_pixels, int C, void* stream);
/* img: [batch, 3, H, W] */
int fo1_nchw_to_hwc8_bf16(const void* img, int is_f32, void* out, int H, int W, int batch, void* stream);
int fo1_gather_rows_bf16(const void* table0, int ld0, const void* table1, int ld1, const void* table2,
                         int ld2, const int32_t* plan, void* out, int ldo, int R, int D, void* stream);

/* Decode-step attention of ONE new token against the KV cache, split over 64-key chunks across workgroups
 * (grid = max chunks x KV heads; the query heads sharing a KV head ride as MFMA query columns), chunk count taken
 * from the device-side kv length so the launch replays inside a hipGraph; partials merged in a fixed order.
 * q: bf16 [n_q_heads*head_dim]; kcache/vtcache as for fo1_attention_bf16; out bf16 [n_q_heads*head_dim].
 * (reference: the 1-token fast path of omchat_qwen2_5_vl.py:143-155 + attention at modeling_qwen2_5_vl.py:738-802) */
size_t fo1_attention_decode_workspace_bytes(int max_kv_len, int n_kv_heads, int head_dim);
int fo1_attention_decode_bf16(const void* q, const void* kcache, long long k_tok_stride, long long k_head_stride,
                              const void* vtcache, long long vt_row_stride, void* out, const int32_t* dyn_kv_len,
                              int max_kv_len, int n_q_heads, int n_kv_heads, int head_dim, float scale,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Batched greedy decode: B <= 32 sequences advance one token per step through ONE stream of the weights, and
 * every position-dependent quantity lives in device memory, so a single captured hipGraph serves every step.
 * Reference: decode fast path omchat_qwen2_5_vl.py:143-155, positions modeling_qwen2_5_vl.py:1848-1860, stop rule
 * mm_utils.py:137-181 + HF greedy search (stop AFTER appending an EOS / keyword id, or at max_new_tokens).
 *   state: int32[B][8] = { pos, rope_row, kv_start, finished, n_gen, max_new, -, - }
 *     pos = cache row the fed token's K row / V^T column is written to; keys attended = [kv_start, pos];
 *     rope_row = row of the [positions, 128] mRoPE tables (cache position + rope delta).
 *   fo1_gemv_batch_bf16            C[M<=32,N] = epilogue(rmsnorm?(x) W^T): mode 0 bias/residual, 1 interleaved SwiGLU,
 *                                  2 fused QKV (bias -> bf16 -> mRoPE -> q rows out, K row + V^T column appended at state.pos)
 *   fo1_attention_decode_batch_bf16  split-KV attention of B one-token queries against their slots
 *   fo1_decode_argmax_accept       greedy pick per logits row + on-device accept: record id, stop check, advance state,
 *                                  (n_stop == -1: PER-SEQUENCE stop sets — stop_ids is a table int32 [sets][17] = {count, ids[16]} and state[b][6] the
 *                                  row of sequence b: a decode pool mixing requests with different stop rules, round 5)
 *                                  next step's embedding-gather plan
 *   fo1_kv_relocate                packed prefill rows -> per-sequence decode slots, all layers in one launch
 * ---------------------------------------------------------------------- */
int fo1_gemv_batch_bf16(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                        void* C, int ldc, int M, int N, int K, int mode, const void* norm_weight, float norm_eps,
                        int n_q_heads, int n_kv_heads, const void* cos_table, const void* sin_table,
                        const int32_t* state, void* kcache, long long kcache_head_stride, void* vtcache,
                        long long vt_row_stride, void* stream);
size_t fo1_attention_decode_batch_workspace_bytes(int max_kv_len, int n_kv_heads, int head_dim, int batch);
int fo1_attention_decode_batch_bf16(const void* q, long long q_seq_stride, const void* kcache, long long k_tok_stride,
                                    long long k_head_stride, const void* vtcache, long long vt_row_stride, void* out,
                                    long long out_seq_stride, const int32_t* state, int batch, int max_kv_len,
                                    int n_q_heads, int n_kv_heads, int head_dim, float scale, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* The split-KV half of fo1_attention_decode_batch_bf16 alone (reference: the 1-token fast path of omchat_qwen2_5_vl.py:143-155 through
 * modeling_qwen2_5_vl.py:738-802): per sequence, KV head and chunk of *kv_chunk_out keys the unnormalised fp32 rows + (max, sum) go to
 * `workspace` as [batch][chunks][n_kv_heads][16][head_dim + 2] (*part_seq_stride_out floats per sequence).  Consumer:
 * fo1_gemv_attn_combine_bf16.  Workspace size: fo1_attention_decode_batch_workspace_bytes. */
int fo1_attention_decode_batch_partials_bf16(const void* q, long long q_seq_stride, const void* kcache, long long k_tok_stride, long long k_head_stride,
                                             const void* vtcache, long long vt_row_stride, const int32_t* state, int batch, int max_kv_len,
                                             int n_q_heads, int n_kv_heads, int head_dim, float scale, void* workspace, size_t workspace_bytes,
                                             int* kv_chunk_out, long long* part_seq_stride_out, void* stream);
/* The o-projection of a decode step at <= 2 sequences with the attention combine in its prologue (modeling_qwen2_5_vl.py:797-800: o_proj of the
 * attention output, + residual at :1075): C[M, N] = bf16(bf16(x W^T) + residual), x = the rows attn_decode_combine would have written from
 * `part` — bit for bit, one shared routine — without that launch and without the [M, heads x 128] activation.  M <= 2, K = n_q_heads * 128 <= 2048,
 * N <= 4096, head_dim 128. */
int fo1_gemv_attn_combine_bf16(const float* part, long long part_seq_stride, const int32_t* state, int kv_chunk, int n_q_heads, int n_kv_heads,
                               const void* W, int ldw, const void* residual, int ldr, void* C, int ldc, int M, int N, void* stream);
int fo1_decode_argmax_accept(const void* logits, long long ld_logits, int n_vocab, int B, const int32_t* first_tokens,
                             int32_t* state, int32_t* plan, int32_t* ids_out, int ids_ld, const int32_t* stop_ids,
                             int n_stop, int32_t* done, void* scratch, void* stream);
int fo1_kv_relocate(const void* ksrc, void* kdst, long long ks_layer, long long ks_head, long long kd_layer,
                    long long kd_head, const void* vsrc, void* vdst, long long vs_layer, long long vs_row,
                    long long vd_layer, long long vd_row, const int32_t* seqs, int B, int max_len, int n_kv_heads,
                    int n_layers, void* stream);

/* ------------------------------------------------------------------------
 * Stage-level entries: one call per fused stage, for hosts that do not want to sequence the primitives
 * themselves.  Every pointer is a device pointer unless marked host; weights are bf16 in the engine's layouts (below); the
 * caller owns all memory including the workspace (sizes from the *_workspace_bytes queries); calls are asynchronous on
 * `stream`, allocate nothing, and may be captured in a hipGraph.  Each entry issues the primitive launches of the
 * Python mirror (vlm_fo1_amd/vit.py, llm.py) — results are bit-identical to it (tests/test_stage_abi_gpu.py); where the mirror takes
 * a fused form (q/k/v epilogue, implicit-GEMM convolution) the LLM, DaViT and SimpleFPN entries take it too, the ViT entry keeps the
 * two-launch q/k/v form (same bits; profiles/r05_stage_abi_path.json).
 *
 *   fo1_vit_forward      Qwen2.5-VL vision tower over packed patch rows (one or several images): window-ordered patch embed,
 *                        `depth` blocks (RMSNorm, QKV, 2-D RoPE, windowed / full attention, proj, SwiGLU MLP), merger; emits the
 *                        merged image tokens and the raster feature map of every full-attention block asked for.
 *                        reference: modeling_qwen2_5_vl.py:436-504 (blocks :306-357, merger :140-158) as driven by
 *                        qwen2_5_vl_encoder.py:37-80,86-158,228-257
 *   fo1_llm_prefill      36-layer Qwen2.5 decoder over packed prompt rows (varlen causal segments), KV cache written at
 *                        [pos0, pos0 + rows); last-row final norm + lm_head + greedy pick per sequence.  Where the q/k/v product runs on the
 *                        256 x 256 GEMM kernel (fo1_gemm_takes_big_tile) and pos0 / the cache pitches are multiples of 8 rows, mRoPE, the K
 *                        append and V^T ride in its epilogue (fo1_qkv_proj_rope_bf16), as in the Python mirror; otherwise GEMM + fo1_qkv_post_llm_bf16.
 *                        reference: modeling_qwen2_5_vl.py:1014-1095,1126-1242; omchat_qwen2_5_vl.py:143-155
 *   fo1_llm_decode_step  one token for `batch` sequences (state / plan / stop rule as in the batched decode block above).
 * ---------------------------------------------------------------------- */
typedef struct fo1_vit_block {   /* bf16 device pointers; Linear weights [out, in] */
    const void* n1; const void* n2;                 /* RMSNorm weights [hidden]                                          */
    const void* wqkv; const void* bqkv;             /* [3 hidden, hidden], [3 hidden]                                    */
    const void* wo; const void* bo;                 /* [hidden, hidden], [hidden]                                        */
    const void* wgu; const void* bgu;               /* gate/up interleaved in 16-row groups, rows padded: [2 ff_padded, hidden] */
    const void* wd; const void* bd;                 /* [hidden, ff_padded] (zero columns beyond ff), [hidden]            */
    const void* wqkv_hm; const void* bqkv_hm;       /* optional (NULL: never fused), ABI 7: the q/k/v weight and bias HEAD-MAJOR, per head [q 80 | k 80 | v 80 |
                                                       16 zero rows] = [256 n_heads, hidden], [256 n_heads] (fo1_qkv_proj_rope_bf16 mode 1; head_dim 80 only):
                                                       passes whose q/k/v product runs on the 256 x 256 GEMM kernel then rotate and transpose in its epilogue */
} fo1_vit_block_t;
typedef struct fo1_vit_weights {
    int32_t depth, hidden, n_heads, ff_padded, k_in, k_in_padded, merge, out_hidden;
    int32_t n_fullatt; int32_t fullatt[8];          /* indices of the full-attention blocks, ascending                    */
    const void* patch_w;                            /* [hidden, k_in_padded] (zero columns beyond k_in)                  */
    const fo1_vit_block_t* blocks;                  /* HOST array [depth]                                                 */
    const void* ln_q; const void* m0w; const void* m0b; const void* m2w; const void* m2b;   /* merger                    */
} fo1_vit_weights_t;
typedef struct fo1_vit_plan {    /* index plan of the packed images (host mirror: vlm_fo1_amd/vit.py GridPlan / BatchPlan) */
    int32_t S;                                      /* patch rows in this pass (multiple of merge^2)                      */
    const int32_t* plan_in;                         /* [S][2]   window-order row r reads input row plan_in[r][1]          */
    const int32_t* plan_raster;                     /* [S][2]   raster row -> window-order row                            */
    const int32_t* plan_tokens;                     /* [S/4][2] raster-merged token -> window-order merge unit            */
    const float* cos; const float* sin;             /* [S][head_dim/2] 2-D rope angles in window order                    */
    const int32_t* items_win; int32_t n_items_win, q_block_win;       /* attention work items (fo1_attention_bf16); q_block_win 0 = every item at most 64 keys: fo1_attention_windows_bf16 */
    const int32_t* items_full; int32_t n_items_full, q_block_full;
    double flops_win, flops_full;                   /* profiler hints                                                     */
} fo1_vit_plan_t;
size_t fo1_vit_workspace_bytes(const fo1_vit_weights_t* w, int S);
int fo1_vit_forward(const fo1_vit_weights_t* w, const fo1_vit_plan_t* plan, const void* pixel_rows /* bf16 [S, ld_pixels] */,
                    int ld_pixels, void* tokens_out /* bf16 [S/4, out_hidden] */,
                    void* const* feature_maps_out /* HOST array [n_fullatt] of bf16 [S, hidden] destinations, NULL entries skipped */,
                    void* workspace, size_t workspace_bytes, void* stream);

typedef struct fo1_llm_layer {
    const void* ln1; const void* ln2;               /* input / post-attention RMSNorm [hidden]                            */
    const void* wqkv; const void* bqkv;             /* [(n_heads + 2 n_kv) head_dim, hidden] rows q | k | v, and bias     */
    const void* wo;                                 /* [hidden, n_heads head_dim]                                         */
    const void* wgu;                                /* gate/up interleaved in 16-row groups [2 intermediate, hidden]      */
    const void* wdown;                              /* [hidden, intermediate]                                             */
} fo1_llm_layer_t;
typedef struct fo1_llm_weights {
    int32_t n_layers, hidden, n_heads, n_kv_heads, head_dim, intermediate, vocab;
    float rms_eps;
    const fo1_llm_layer_t* layers;                  /* HOST array [n_layers]                                              */
    const void* embed; const void* final_norm; const void* lm_head;    /* [vocab, hidden], [hidden], [vocab, hidden]      */
} fo1_llm_weights_t;
typedef struct fo1_kv_cache {    /* K [layer][kv_head][row][head_dim]; V^T [layer][kv_head * head_dim][row]               */
    void* k; long long k_layer_stride, k_head_stride;                  /* elements                                        */
    void* vt; long long vt_layer_stride, vt_row_stride;
    int32_t capacity;                                                  /* rows                                            */
} fo1_kv_cache_t;
size_t fo1_llm_prefill_workspace_bytes(const fo1_llm_weights_t* w, int rows, int n_seq);
int fo1_llm_prefill(const fo1_llm_weights_t* w, const fo1_kv_cache_t* kv,
                    const void* embeds /* bf16 [rows, ld_embeds]: spliced prompt rows */, int ld_embeds,
                    const void* cos, const void* sin /* bf16 [rows, head_dim] section-selected mRoPE tables */,
                    int rows, int pos0,
                    const int32_t* items, int n_items, int q_block, double attn_flops /* causal segments, fo1_attention_bf16 */,
                    const int32_t* last_plan /* [n_seq][2] = {0, last row of sequence b} */, int n_seq,
                    void* hidden_out /* optional bf16 [rows, hidden]: final residual stream */,
                    void* last_hidden /* bf16 [n_seq, hidden] after the final norm */, void* logits /* bf16 [n_seq, vocab] */,
                    int32_t* next_ids /* [n_seq] */, void* workspace, size_t workspace_bytes, void* stream);
size_t fo1_llm_decode_workspace_bytes(const fo1_llm_weights_t* w, int batch, int slot_rows);
int fo1_llm_decode_step(const fo1_llm_weights_t* w, const fo1_kv_cache_t* slots, const void* rope_cos, const void* rope_sin,
                        int32_t* state, int32_t* plan, int32_t* ids_out, int ids_ld, const int32_t* stop_ids, int n_stop,
                        int32_t* done, int batch, int slot_rows, int max_kv_len /* bound on any sequence's keys this step, <= slot_rows:
                        picks the attention geometry */, void* logits /* bf16 [batch, vocab] */, void* workspace,
                        size_t workspace_bytes, void* stream);
/*   fo1_davit_forward      DaViT-L aux tower over `batch` same-size images: 4 stages of ConvEmbed + (SpatialBlock, ChannelBlock)
 *                          pairs; emits the four token-major stage maps the HFRE reads.  reference: davit_aux_encoder.py:54-69,
 *                          davit/modeling_davit.py:102-148 (ConvEmbed), :175-205 (ChannelBlock), :284-330 (SpatialBlock), :478-506
 *   fo1_simplefpn_forward  ViTDet SimpleFPN on the last ViT map (strides 3.5 / 7 / 14 / 28).  reference: simple_fpn.py:100-216
 *   fo1_projector_forward  mlpN_gelu connector (mm_projector / mm_projector_aux).  reference: multimodal_projector/builder.py:64-71,103-110 */
typedef struct fo1_davit_half {   /* one SpatialBlock or ChannelBlock; bf16 device pointers                              */
    const void* conv1_w; const void* conv1_b; const void* conv2_w; const void* conv2_b;   /* depthwise 3x3, tap-major [9, C], [C] */
    const void* an_w; const void* an_b;           /* attention pre-norm (LayerNorm)                                      */
    const void* qkv_w; const void* qkv_b; const void* proj_w; const void* proj_b;
    const void* fn_w; const void* fn_b;           /* FFN pre-norm                                                        */
    const void* fc1_w; const void* fc1_b; const void* fc2_w; const void* fc2_b;           /* [4C, C], [C, 4C]             */
} fo1_davit_half_t;
typedef struct fo1_davit_block { fo1_davit_half_t spatial, channel; } fo1_davit_block_t;
typedef struct fo1_davit_stage {
    int32_t dim, heads, depth, kernel, stride, pad, prenorm, K_padded;   /* ConvEmbed geometry; K_padded = GEMM K (64-multiple) */
    const void* conv_w; const void* conv_b;       /* [dim, K_padded] rows (ky, kx, cin), zero pad columns; [dim]         */
    const void* norm_w; const void* norm_b;       /* ConvEmbed LayerNorm (pre: over the input channels; post: over dim)  */
    const fo1_davit_block_t* blocks;              /* HOST array [depth]                                                  */
} fo1_davit_stage_t;
typedef struct fo1_davit_weights { int32_t n_stages, window; fo1_davit_stage_t stages[4]; } fo1_davit_weights_t;
typedef struct fo1_davit_plan {    /* per image size: window-attention work items of every stage (fo1_attention_bf16)     */
    int32_t H, W, batch;
    const int32_t* items[4]; int32_t n_items[4]; int32_t q_block[4];
} fo1_davit_plan_t;
size_t fo1_davit_workspace_bytes(const fo1_davit_weights_t* w, const fo1_davit_plan_t* plan);
int fo1_davit_forward(const fo1_davit_weights_t* w, const fo1_davit_plan_t* plan, const void* images /* [batch, 3, H, W] bf16 or fp32 */,
                      int images_are_f32, void* const* maps_out /* HOST array [4]: bf16 [batch * H_i * W_i, dim_i] */,
                      void* workspace, size_t workspace_bytes, void* stream);

typedef struct fo1_fpn_head { const void* w1; const void* n1_w; const void* n1_b; const void* w3; const void* n3_w; const void* n3_b; } fo1_fpn_head_t;
typedef struct fo1_fpn_weights {
    int32_t c_in, c_up1, c_up2, c_out;            /* 1280, 640, 320, 512                                                 */
    const void* t1a_w; const void* t1a_b;         /* ConvTranspose2d as GEMM rows (dy, dx, co): [4 c_up1, c_in], bias x4 */
    const void* t1_ln_w; const void* t1_ln_b;
    const void* t1b_w; const void* t1b_b;         /* [4 c_up2, c_up1]                                                    */
    const void* t2_w; const void* t2_b;           /* [4 c_up1, c_in]                                                     */
    fo1_fpn_head_t heads[4];                      /* 1x1 conv [c_out, C_level], LN, 3x3 conv [c_out, 9 c_out], LN        */
} fo1_fpn_weights_t;
size_t fo1_simplefpn_workspace_bytes(const fo1_fpn_weights_t* w, int H, int W, int batch);
int fo1_simplefpn_forward(const fo1_fpn_weights_t* w, const void* vit_map /* bf16 [batch*H*W, c_in] raster */, int H, int W, int batch,
                          void* const* maps_out /* HOST array [4]: (4H,4W), (2H,2W), (H,W), (H/2,W/2) x c_out */, void* workspace,
                          size_t workspace_bytes, void* stream);

typedef struct fo1_projector { int32_t n_layers; int32_t dims[5]; const void* w[4]; const void* b[4]; } fo1_projector_t;
size_t fo1_projector_workspace_bytes(const fo1_projector_t* p, int rows);
int fo1_projector_forward(const fo1_projector_t* p, const void* x, int ldx, int rows, void* out, int ld_out, void* workspace,
                          size_t workspace_bytes, void* stream);
/* Zero-fill as a kernel launch (graph-capture safe); p 16-byte aligned, bytes % 16 == 0. */
int fo1_zero_bytes(void* p, size_t bytes, void* stream);

/* ------------------------------------------------------------------------
 * Image preprocessing, device side  
 * Replaces, after the host's PIL decode + bicubic resize, the rescale / normalise / layout work of
 *   Qwen2VLImageProcessor (qwen2_5_vl_encoder.py:206-212 -> pixel_values [S, 1176], patches in 2x2 merge-block order,
 *   vector (C=3, T=2, 14, 14), frame duplicated along T)   -> fo1_patchify_u8_bf16
 *   CLIPImageProcessor.preprocess (image_processing_clip.py:222-367; davit/configs.py:139-152) -> fo1_normalize_u8_bf16
 * image: uint8 [H, W, 3] (RGB, HWC) on the device; lut: bf16 [3][256] = bf16((v/255 - mean_c)/std_c) built by the host with
 * the reference's arithmetic, so outputs are bit-identical to "CPU processor then .to(bfloat16)".
 * ---------------------------------------------------------------------- */
int fo1_patchify_u8_bf16(const void* image_hwc_u8, int H, int W, const void* lut_bf16, void* out, int ld,
                         int patch, int merge, void* stream);
int fo1_normalize_u8_bf16(const void* image_hwc_u8, int H, int W, const void* lut_bf16, void* out_chw, void* stream);

/* ------------------------------------------------------------------------
 * Multi-scale deformable attention, forward  (the UPN proposal detector's operator, the reference's only
 * native code).  Replaces MSDA.ms_deform_attn_forward (detect_tools/upn/ops/src/ms_deform_attn.h:21-36 ->
 * ops/src/cuda/ms_deform_attn_cuda.cu:25-80 -> ms_deformable_im2col_gpu_kernel, ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299,
 * bilinear helper :32-84), called by MSDeformAttnFunction.forward (ops/functions/ms_deform_attn_func.py:23-28) from
 * MSDeformAttn.forward (ops/modules/ms_deform_attn.py:186-202).  The reference's `im2col_step` only batches its launches and
 * has no counterpart.  Backward (training) is not built: the reference path is inference.
 *   value [N, S, M, D] (S = sum H_l W_l, level-major); spatial_shapes int64 [L, 2] = (H_l, W_l) and level_start_index
 *   int64 [L], both ON THE DEVICE as the reference passes them; sampling_loc [N, Lq, M, L, P, 2] = (x, y), the unit square is
 *   the map, zero padding outside; attn_weight [N, Lq, M, L, P]; out [N, Lq, M * D].  All contiguous.
 *   dtype 0: everything fp32 (the reference's default);  1: everything fp64 (the reference test's double check);
 *   2: value / out bf16, sampling_loc / attn_weight fp32 (engine form).  fp32 (fp64 for dtype 1) accumulation in the
 *   reference's operation order. */
int fo1_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const void* sampling_loc, const void* attn_weight, int N, int S, int M, int D, int L, int Lq,
                               int P, void* out, int dtype, void* stream);

/* Fused MSDeformAttn core (the UPN encoder / decoder layers' attention between their Linear layers, ops/modules/ms_deform_attn.py:
 * 133-202): softmax over each head's L*P logits, sampling locations from the reference points and raw offsets, bilinear gather and
 * weighted sum in ONE launch — the [N,Lq,M,L,P,2] sampling_locations and [N,Lq,M,L,P] attention_weights tensors never exist.
 *   value bf16 [N, S, M*D] (value_proj output); offsets_logits fp32 [N, Lq, M*L*P*3] = one GEMM whose weight rows are
 *   [sampling_offsets (M*L*P*2) | attention_weights (M*L*P)]; reference_points fp32 [N, Lq, ref_levels, ref_dim], ref_dim 2 (points:
 *   loc = ref + off / (W_l, H_l), :150-157) or 4 (boxes cx, cy, w, h: loc = ref[:2] + off / P * ref[2:] * 0.5, :169-175);
 *   out bf16 [N, Lq, M*D] (input of output_proj).  D % 8 == 0. */
int fo1_msda_fused_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                        const float* offsets_logits, const float* reference_points, int ref_levels /* L, or 1 = same at every level */,
                        int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, void* out, void* stream);
/* y = bf16(a + b) over [M, D] bf16 rows (with_pos_embed of the DETR-style layers, encoder/upn_encoder.py:62-63). D % 8 == 0. */
int fo1_add_bf16(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int M, int D, void* stream);

/* ------------------------------------------------------------------------
 * UPN proposal detector, Swin-L backbone pieces (detect_tools/upn/models/backbone/swin.py).
 *   fo1_attention_window_bias_bf16   W-MSA / SW-MSA (:136-175): fo1_attention_bf16 over windows + fp32 bias [heads][wlen][wlen]
 *                                    (the relative-position bias gathered per layer at load) + the shifted-window mask (-100
 *                                    between shift regions, BasicLayer.forward :446-466) computed from the window's place
 *   fo1_swin_window_partition_bf16   pad to the window multiple (zeros, AFTER norm1 as the reference), cyclic shift (torch.roll by
 *                                    -shift), window_partition (:42-55, :273-296) in one gather; xw [batch*nW*ws*ws, C]
 *   fo1_swin_window_reverse_add_bf16 window_reverse + inverse shift + crop + residual add (:299-313)
 *   fo1_patch_merge_bf16             PatchMerging's gather [x(2i,2j) | x(2i+1,2j) | x(2i,2j+1) | x(2i+1,2j+1)], zero beyond an odd edge
 *                                    (:333-352); out [batch*ceil(H/2)*ceil(W/2), 4C]
 *   fo1_groupnorm_tokens_bf16        nn.GroupNorm(groups, C) of input_proj (architecture/upn_model.py:246-262) on a token-major map
 * ---------------------------------------------------------------------- */
int fo1_attention_window_bias_bf16(const void* Q, long long q_tok_stride, long long q_head_stride, const void* K, long long k_tok_stride,
                                   long long k_head_stride, const void* VT, long long vt_row_stride, void* O, long long o_tok_stride,
                                   long long o_head_stride, const int32_t* items, int n_items, int q_block, int n_heads, int head_dim,
                                   float scale, const float* bias, int wlen, int ws, int shift, int nwy, int nwx, double flops_hint,
                                   void* stream);
int fo1_swin_window_partition_bf16(const void* x, void* xw, int H, int W, int C, int ws, int shift, int batch, void* stream);
int fo1_swin_window_reverse_add_bf16(const void* yw, const void* shortcut, void* y, int H, int W, int C, int ws, int shift, int batch,
                                     void* stream);
int fo1_patch_merge_bf16(const void* x, void* out, int H, int W, int C, int batch, void* stream);
size_t fo1_groupnorm_tokens_workspace_bytes(int S, int groups);
int fo1_groupnorm_tokens_bf16(const void* x, int ldx, int S, int C, int groups, const void* weight, const void* bias, float eps, void* y,
                              int ldy, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * UPN proposal detector, query selection and decoder helpers (detect_tools/upn/models/...).
 *   fo1_sine_embed_bf16   gen_sineembed_for_position (utils/detr_utils.py:276-310): ref fp32 [n, dims] (x, y[, w, h]) ->
 *                         bf16 [n, dims*128], 128-wide blocks ordered (y, x[, w, h]), temperature 10000, scale 2 pi
 *   fo1_box_refine_f32    mode 0: sigmoid(delta + inverse_sigmoid(ref)) (decoder/upn_decoder.py:336-341, architecture/upn_model.py:
 *                         110-117; inverse_sigmoid eps 1e-3, utils/detr_utils.py:269-273);  mode 1: delta + ref (encoder
 *                         proposals in logit space, deformable_transformer.py:299-301);  mode 2: sigmoid(delta + ref) (ref in logit
 *                         space: the decoder's first reference points, upn_decoder.py:290).  [n, 4] fp32 with row strides
 *   fo1_mask_rows_bf16    y[m] = keep[m] ? x[m] : 0 (gen_encoder_output_proposals zeroes invalid tokens' memory, detr_utils.py:402-406)
 *   fo1_topk_desc_f32     torch.topk(scores, k) of deformable_transformer.py:305: indices (and values) of the k largest of
 *                         scores[i*stride], descending, ties -> lower index; single-workgroup bitonic sort in `workspace`
 *   fo1_gather_rows_f32   out[i] = table[idx[i]] (fp32 rows; torch.gather of the selected proposals, :311-316)
 * ---------------------------------------------------------------------- */
int fo1_sine_embed_bf16(const float* ref, int ld_ref, int n, int dims, void* out, int ld_out, void* stream);
int fo1_box_refine_f32(const float* delta, int ld_delta, const float* ref, int ld_ref, float* out, int ld_out, int n, int mode, void* stream);
int fo1_mask_rows_bf16(const void* x, int ldx, const uint8_t* keep, void* y, int ldy, int M, int D, void* stream);
size_t fo1_topk_workspace_bytes(int n);
int fo1_topk_desc_f32(const float* scores, int stride, int n, int k, int32_t* idx_out, float* val_out, void* workspace,
                      size_t workspace_bytes, void* stream);
int fo1_gather_rows_f32(const float* table, int ld_table, const int32_t* idx, float* out, int ld_out, int n, int D, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FO1_H */
