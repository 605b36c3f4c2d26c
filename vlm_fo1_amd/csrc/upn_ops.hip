// upn_ops.hip — small kernels of the UPN proposal detector's query selection and decoder for gfx950.
// Reference: detect_tools/upn/models/utils/detr_utils.py (inverse_sigmoid :269-273, gen_sineembed_for_position :276-310),
// models/architecture/deformable_transformer.py (get_two_stage_proposal :262-336: torch.topk over the per-token scores),
// models/decoder/upn_decoder.py (box refinement :336-341), models/architecture/upn_model.py (:110-117).
// All tiny (900 queries, ~22k tokens): latency, not throughput; what matters is that they keep the forward on the device
// (no host round trip between the encoder and the decoder) and graph-capturable.
#include "common.h"

namespace fo1 {

// ---- sine embedding of reference boxes: [n, dims] (x, y[, w, h]) fp32 -> bf16 [n, dims*128], blocks ordered (y, x[, w, h]) ----
__global__ __launch_bounds__(256) void sine_embed_kernel(const float* __restrict__ ref, int ld_ref, int n, int dims, uint16_t* __restrict__ out, int ldo) {
    const int per = dims * 128;
    const long long total = (long long)n * per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / per), c = (int)(i - (long long)row * per);
        const int blk = c >> 7, j = c & 127;
        const int src = blk == 0 ? 1 : (blk == 1 ? 0 : blk);          // block 0 = y, block 1 = x, then w, h
        const float v = ref[(long long)row * ld_ref + src] * 6.283185307179586f;
        const float dim_t = powf(10000.0f, (float)(2 * (j / 2)) / 128.0f);
        const float a = v / dim_t;
        out[(long long)row * ldo + c] = f32_to_bf16((j & 1) ? cosf(a) : sinf(a));
    }
}

// ---- box arithmetic in logit space -------------------------------------------------------------------------------------------
// mode 0: out = sigmoid(delta + inverse_sigmoid(ref))   (decoder refinement / final boxes; eps 1e-3 as the reference)
// mode 1: out = delta + ref                             (encoder proposals: ref already holds logits, +inf for invalid tokens)
// mode 2: out = sigmoid(delta + ref)                    (ref in logit space)
__global__ __launch_bounds__(256) void box_refine_kernel(const float* __restrict__ delta, int ld_delta, const float* __restrict__ ref, int ld_ref,
                                                         float* __restrict__ out, int ld_out, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4) return;
    const int row = i >> 2, c = i & 3;
    const float d = delta[(long long)row * ld_delta + c];
    float r = ref[(long long)row * ld_ref + c];
    if (mode == 0) {
        float x = fminf(fmaxf(r, 0.0f), 1.0f);
        const float x1 = fmaxf(x, 1e-3f), x2 = fmaxf(1.0f - x, 1e-3f);
        const float u = d + logf(x1 / x2);
        out[(long long)row * ld_out + c] = 1.0f / (1.0f + expf(-u));
    } else if (mode == 1) {
        out[(long long)row * ld_out + c] = d + r;
    } else {
        out[(long long)row * ld_out + c] = 1.0f / (1.0f + expf(-(d + r)));
    }
}

// ---- y[m, :] = keep[m] ? x[m, :] : 0 (gen_encoder_output_proposals zeroes the memory rows of invalid proposals, :402-406) ----
__global__ __launch_bounds__(256) void mask_rows_kernel(const uint16_t* __restrict__ x, int ldx, const uint8_t* __restrict__ keep, uint16_t* __restrict__ y,
                                                        int ldy, int M, int D) {
    const int chunks = D >> 3;
    const long long total = (long long)M * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i - (long long)m * chunks);
        uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)m * ldx + c * 8);
        if (!keep[m]) v = uint4{0, 0, 0, 0};
        *reinterpret_cast<uint4*>(y + (size_t)m * ldy + c * 8) = v;
    }
}

// ---- top-k (descending, ties -> lower index first) by a single-workgroup bitonic sort of 64-bit (key, ~index) words ----------
// n <= 2^17 tokens; one launch, no host round trip.  key = order-preserving map of the fp32 score (NaN sorts last).
__device__ __forceinline__ unsigned int f32_order(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;             // NaN: smallest
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void topk_bitonic_kernel(const float* __restrict__ scores, int stride, int n, int n_pad, int k,
                                                            unsigned long long* __restrict__ ws, int* __restrict__ idx_out, float* __restrict__ val_out) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n_pad; i += 1024) {
        unsigned long long w = 0ull;                              // padding sorts last (key 0 is below every real key; NaN rows share it)
        if (i < n) w = ((unsigned long long)f32_order(scores[(long long)i * stride]) << 32) | (unsigned int)(~(unsigned int)i);
        ws[i] = w;
    }
    __syncthreads();
    // descending bitonic sort
    for (int kk = 2; kk <= n_pad; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pad; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = ws[i], b = ws[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (a < b) : (a > b)) { ws[i] = b; ws[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 1024) {
        const unsigned long long w = ws[i];
        const int idx = (int)(~(unsigned int)(w & 0xffffffffull));
        idx_out[i] = idx;
        if (val_out) val_out[i] = scores[(long long)idx * stride];
    }
}

// ---- out[i, :] = table[idx[i], :]  fp32 rows (gather of the selected proposals' coordinates) --------------------------------
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ table, int ldt, const int* __restrict__ idx, float* __restrict__ out,
                                                              int ldo, int n, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int row = i / D, c = i - row * D;
    out[(long long)row * ldo + c] = table[(long long)idx[row] * ldt + c];
}

}  // namespace fo1

extern "C" {

int fo1_sine_embed_bf16(const float* ref, int ld_ref, int n, int dims, void* out, int ld_out, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(ref && out && n > 0 && (dims == 2 || dims == 4) && ld_ref >= dims && ld_out >= dims * 128, "sine_embed: bad arguments");
    const long long total = (long long)n * dims * 128;
    FO1_LAUNCH("sine_embed", (double)total * 2.0, sine_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref, ld_ref, n,
               dims, (uint16_t*)out, ld_out);
    return FO1_OK;
}

int fo1_box_refine_f32(const float* delta, int ld_delta, const float* ref, int ld_ref, float* out, int ld_out, int n, int mode, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(delta && ref && out && n > 0 && ld_delta >= 4 && ld_ref >= 4 && ld_out >= 4 && mode >= 0 && mode <= 2, "box_refine: bad arguments");
    FO1_LAUNCH("box_refine", (double)n * 48.0, box_refine_kernel, dim3((n * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, delta, ld_delta, ref, ld_ref, out,
               ld_out, n, mode);
    return FO1_OK;
}

int fo1_mask_rows_bf16(const void* x, int ldx, const uint8_t* keep, void* y, int ldy, int M, int D, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && keep && y && D > 0 && D % 8 == 0 && ldx >= D && ldy >= D && ldx % 8 == 0 && ldy % 8 == 0, "mask_rows: bad arguments");
    if (M == 0) return FO1_OK;
    const long long total = (long long)M * (D / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    FO1_LAUNCH("mask_rows", (double)M * D * 4.0, mask_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx, keep, (uint16_t*)y,
               ldy, M, D);
    return FO1_OK;
}

size_t fo1_topk_workspace_bytes(int n) {
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    return (size_t)n_pad * sizeof(unsigned long long);
}

// idx_out[0..k) = indices of the k largest scores[i * stride], i < n, in descending score order (ties: lower index first — torch.topk
// leaves the tie order unspecified); val_out (optional) their values.  One workgroup; n <= 131072.
int fo1_topk_desc_f32(const float* scores, int stride, int n, int k, int32_t* idx_out, float* val_out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(scores && idx_out && workspace && n > 0 && n <= 131072 && k > 0 && k <= n && stride >= 1, "topk: bad arguments (n=%d k=%d)", n, k);
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    if (workspace_bytes < (size_t)n_pad * 8) return set_err(FO1_ERR_WORKSPACE, "topk: workspace %zu B < %zu B", workspace_bytes, (size_t)n_pad * 8);
    FO1_LAUNCH("topk_bitonic", (double)n * 4.0, topk_bitonic_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, stride, n, n_pad, k,
               (unsigned long long*)workspace, (int*)idx_out, val_out);
    return FO1_OK;
}

int fo1_gather_rows_f32(const float* table, int ld_table, const int32_t* idx, float* out, int ld_out, int n, int D, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(table && idx && out && n > 0 && D > 0 && ld_table >= D && ld_out >= D, "gather_rows_f32: bad arguments");
    FO1_LAUNCH("gather_rows_f32", (double)n * D * 8.0, gather_rows_f32_kernel, dim3((n * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, ld_table,
               (const int*)idx, out, ld_out, n, D);
    return FO1_OK;
}

}  // extern "C"
