// decode_common.h — types shared by the decode-step translation units (decode.hip: v_dot2 GEMV, bookkeeping; decode_mfma.hip:
// MFMA skinny GEMV; attention.hip: decode attention).  Device state per sequence: int32[8] =
// { pos, rope_row, kv_start, finished, n_gen, max_new, -, - } (see decode.hip).
#pragma once
#include "common.h"

namespace fo1 {

struct GemvBParams {
    const uint16_t* X;       // [M, ldx]
    const uint16_t* W;       // [N, ldw]
    const uint16_t* bias;    // [N] or null
    const uint16_t* res;     // [M, ldr] or null (plain mode)
    uint16_t* C;             // [M, ldc]: plain out | SwiGLU out | rotated q rows (QKV mode)
    int M, N, K, ldx, ldw, ldc, ldr;
    const uint16_t* norm_w;  // optional fused RMSNorm on x
    float norm_eps;
    int kp_chunks;           // (dot2 kernel) 16-B chunks of K staged in LDS at a time
    int canon_chunks;        // (dot2 kernel) canonical K segment (chunks)
    // QKV mode
    int n_q, n_kv;           // heads (head_dim 128)
    const uint16_t* cos_t; const uint16_t* sin_t;   // [rows, 128] bf16 tables
    const int* state;        // [M][8]
    uint16_t* kcache; long long kc_head_stride;      // [n_kv][rows][128]
    uint16_t* vtcache; long long vt_row_stride;      // [n_kv*128][rows]
    // x = the decode attention's output, combined HERE from its split-KV partials (round 6: M <= 2, plain mode, K = heads x 128 <= 2048):
    // the rows attn_decode_combine_kernel would write never exist and its launch is not made
    const float* attn_part = nullptr;                // [M][chunks][n_kv][16][130] fp32 (attn_fwd_kernel PARTIAL)
    long long attn_part_seq_stride = 0;
    const int* attn_state = nullptr;                 // [M][8]: keys = state[0] + 1 - state[2], state[3] = finished
    int attn_chunk = 0, attn_n_kv = 0, attn_group = 0;
};

// One (query head, 8 consecutive head-dim elements) of the split-KV combine: out[d] = sum_s exp(m_s - M) O_s[d] / sum_s exp(m_s - M) l_s over
// the chunks in ascending order — THE arithmetic of attn_decode_combine_kernel (attention.hip), which calls it with ND = 1; the decode GEMV's
// fused prologue calls it with ND = 8.  Explicit fmaf: both callers round identically whatever the contraction setting.
//   pr0 = the (head's slot) row of chunk 0, cstride = floats between consecutive chunks' rows; m_s / l_s at pr[128], pr[129].
template <int ND>
__device__ __forceinline__ void attn_combine_row(const float* __restrict__ pr0, long long cstride, int n_valid, int d0, float (&out)[ND]) {
    // loads in batches, every address of a batch known up front (a load inside `for s` is load -> wait -> use per chunk: ~0.5 us each, see the
    // round-2 note in attention.hip); indices past the last chunk are clamped and enter with weight exactly 0 (fmaf(0, finite, x) = x)
    float M = -INFINITY;
    for (int s0 = 0; s0 < n_valid; s0 += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = pr0[(long long)min(s0 + j, n_valid - 1) * cstride + 128];
#pragma unroll
        for (int j = 0; j < 8; ++j) M = fmaxf(M, t[j]);
    }
    float num[ND], den = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) num[j] = 0.f;
    for (int s0 = 0; s0 < n_valid; s0 += 4) {
        float mm[4], ll[4], v[4][ND];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* pr = pr0 + (long long)min(s0 + u, n_valid - 1) * cstride;
            mm[u] = pr[128];
            ll[u] = pr[129];
#pragma unroll
            for (int j = 0; j < ND; ++j) v[u][j] = pr[d0 + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float w = s0 + u < n_valid ? __expf(mm[u] - M) : 0.f;
#pragma unroll
            for (int j = 0; j < ND; ++j) num[j] = __builtin_fmaf(w, v[u][j], num[j]);
            den = __builtin_fmaf(w, ll[u], den);
        }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) out[j] = den > 0.f ? num[j] / den : 0.f;
}

enum { GB_PLAIN = 0, GB_SWIGLU = 1, GB_QKV = 2 };

// decode_mfma.hip
#ifdef FO1_ENABLE_AB
extern int g_gemv_half;   // decode_mfma.hip: bit 0 = 8-row units at M <= 8 (HALF), bit 1 = at 9..32 sequences (R8)
#else
[[maybe_unused]] static constexpr int g_gemv_half = 3;
#endif
int gemv_mfma_any(GemvBParams& p, int mode, hipStream_t st);

}  // namespace fo1
