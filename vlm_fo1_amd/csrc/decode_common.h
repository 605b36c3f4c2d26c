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
};

enum { GB_PLAIN = 0, GB_SWIGLU = 1, GB_QKV = 2 };

// decode_mfma.hip
#ifdef FO1_ENABLE_AB
extern int g_gemv_half;   // decode_mfma.hip: bit 0 = 8-row units at M <= 8 (HALF), bit 1 = at 9..32 sequences (R8)
#else
[[maybe_unused]] static constexpr int g_gemv_half = 3;
#endif
int gemv_mfma_any(GemvBParams& p, int mode, hipStream_t st);

}  // namespace fo1
