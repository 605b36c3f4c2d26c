// rope.hip — rotary position embedding on the fused QKV activations, and the V -> V^T copy the
// attention kernel consumes.  HBM-bound, 16-byte accesses.
//
//  * LLM mRoPE (reference modeling_qwen2_5_vl.py:603-624,643-685): cos/sin arrive already
//    section-selected ([L,128], bf16 like `cos.to(dtype=x.dtype)` :624); rotate-half form with the
//    reference's three bf16 roundings: bf16( bf16(x*cos) + bf16(rot(x)*sin) ).
//  * ViT 2-D RoPE (:162-169 / :219-230): fp32 math on fp32 cos/sin [S, HD/2], one rounding.
//  * transpose: dst[c, col0 + m] = src[m, c]  (V^T rows = kv_head*HD + d; KV-cache append when col0 > 0)
#include "common.h"

namespace fo1 {

__device__ __forceinline__ void unpack8r(const uint4& u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8r(const float (&f)[8]) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    return u;
}
__device__ __forceinline__ float rb(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// x: [L, ld] rows; heads [0, n_heads) of width HD starting at column col0 are rotated in place
// (n_heads counts q heads + k heads when they are adjacent).  cos/sin: bf16 [L, HD].
// Optionally the rotated K heads are also copied to kcache[kv_head][pos0 + t][HD].
template <int HD>
__device__ __forceinline__ void rope_llm_body(uint16_t* __restrict__ x, int ld, int col0, int n_heads,
                                              const uint16_t* __restrict__ cosb, const uint16_t* __restrict__ sinb, int L,
                                              uint16_t* __restrict__ kcache, int k_first_head, long long kc_head_stride,
                                              int pos0, const int* __restrict__ dyn, int bid, int nblk) {
    constexpr int HC = HD / 16;  // chunk pairs per head (each thread does chunk c and its partner c + HD/16)
    // decode graphs: position and rope-table row come from device memory (dyn = {cache position, table row})
    int row0 = 0;
    if (dyn) { pos0 = dyn[0]; row0 = dyn[1]; }
    const long long total = (long long)L * n_heads * HC;
    for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblk * blockDim.x) {
        const int c = (int)(i % HC);
        const long long r = i / HC;
        const int hd = (int)(r % n_heads), t = (int)(r / n_heads);
        uint16_t* p = x + (long long)t * ld + col0 + hd * HD;
        const int d0 = c * 8;
        float a[8], b[8], ca[8], sa[8], cb[8], sb[8], oa[8], ob[8];
        unpack8r(*reinterpret_cast<const uint4*>(p + d0), a);             // x[d],        d <  HD/2
        unpack8r(*reinterpret_cast<const uint4*>(p + d0 + HD / 2), b);    // x[d + HD/2]
        unpack8r(*reinterpret_cast<const uint4*>(cosb + (long long)(row0 + t) * HD + d0), ca);
        unpack8r(*reinterpret_cast<const uint4*>(sinb + (long long)(row0 + t) * HD + d0), sa);
        unpack8r(*reinterpret_cast<const uint4*>(cosb + (long long)(row0 + t) * HD + d0 + HD / 2), cb);
        unpack8r(*reinterpret_cast<const uint4*>(sinb + (long long)(row0 + t) * HD + d0 + HD / 2), sb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            oa[j] = rb(a[j] * ca[j]) + rb(-b[j] * sa[j]);  // rotate_half: first half pairs with -x2
            ob[j] = rb(b[j] * cb[j]) + rb(a[j] * sb[j]);   //              second half pairs with x1
        }
        const uint4 ua = pack8r(oa), ub = pack8r(ob);
        *reinterpret_cast<uint4*>(p + d0) = ua;
        *reinterpret_cast<uint4*>(p + d0 + HD / 2) = ub;
        if (kcache && hd >= k_first_head) {
            uint16_t* kc = kcache + (long long)(hd - k_first_head) * kc_head_stride + (long long)(pos0 + t) * HD;
            *reinterpret_cast<uint4*>(kc + d0) = ua;
            *reinterpret_cast<uint4*>(kc + d0 + HD / 2) = ub;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void rope_llm_kernel(uint16_t* __restrict__ x, int ld, int col0, int n_heads,
                                                       const uint16_t* __restrict__ cosb, const uint16_t* __restrict__ sinb, int L,
                                                       uint16_t* __restrict__ kcache, int k_first_head, long long kc_head_stride,
                                                       int pos0, const int* __restrict__ dyn) {
    rope_llm_body<HD>(x, ld, col0, n_heads, cosb, sinb, L, kcache, k_first_head, kc_head_stride, pos0, dyn, blockIdx.x, gridDim.x);
}

// ViT: qkv [S, ld]; q heads at col 0, k heads at col n_heads*HD; cos/sin fp32 [S, HD/2]
template <int HD>
__device__ __forceinline__ void rope_vit_body(uint16_t* __restrict__ x, int ld, int n_heads2, const float* __restrict__ cosf_,
                                              const float* __restrict__ sinf_, int S, int bid, int nblk) {
    constexpr int HH = HD / 2, HC = HH / 8;  // 40 -> 5 chunk pairs per head
    const long long total = (long long)S * n_heads2 * HC;
    for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nblk * blockDim.x) {
        const int c = (int)(i % HC);
        const long long r = i / HC;
        const int hd = (int)(r % n_heads2), t = (int)(r / n_heads2);
        uint16_t* p = x + (long long)t * ld + hd * HD;
        const int d0 = c * 8;
        float a[8], b[8], oa[8], ob[8];
        unpack8r(*reinterpret_cast<const uint4*>(p + d0), a);
        unpack8r(*reinterpret_cast<const uint4*>(p + d0 + HH), b);
        const float* cp = cosf_ + (long long)t * HH + d0;
        const float* sp = sinf_ + (long long)t * HH + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cs = cp[j], sn = sp[j];
            // explicit contraction (one product rounded to fp32, the other fused): the q/k/v GEMM's fused epilogue (gemm.hip, epilogue32_qkv)
            // writes the same two expressions and must give the same bits, whatever -ffp-contract would pick for `a * cs - b * sn`
            oa[j] = __builtin_fmaf(a[j], cs, -(b[j] * sn));
            ob[j] = __builtin_fmaf(b[j], cs, a[j] * sn);
        }
        *reinterpret_cast<uint4*>(p + d0) = pack8r(oa);
        *reinterpret_cast<uint4*>(p + d0 + HH) = pack8r(ob);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void rope_vit_kernel(uint16_t* __restrict__ x, int ld, int n_heads2,
                                                       const float* __restrict__ cosf_, const float* __restrict__ sinf_, int S) {
    rope_vit_body<HD>(x, ld, n_heads2, cosf_, sinf_, S, blockIdx.x, gridDim.x);
}

// dst[c * ldd + col0 + m] = src[m * lds + c], tile 64 rows x 64 cols through LDS
__device__ __forceinline__ void transpose_body(const uint16_t* __restrict__ src, int lds_, uint16_t* __restrict__ dst,
                                               long long ldd, int col0, int M, int C, int bx, int by) {
    __shared__ uint16_t tile[64][66];
    const int m0 = bx * 64, c0 = by * 64;
    const int tid = threadIdx.x;
    // load: 64 rows x 8 chunks of 8 channels
    for (int q = tid; q < 64 * 8; q += 256) {
        const int r = q >> 3, cc = q & 7;
        uint4 v = uint4{0, 0, 0, 0};
        if (m0 + r < M) v = *reinterpret_cast<const uint4*>(src + (long long)(m0 + r) * lds_ + c0 + cc * 8);
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[r][cc * 8 + j] = e[j];
    }
    __syncthreads();
    // store: 64 channels x 64 rows; thread -> (channel = q >> 2 .. , 16-row group)
    for (int q = tid; q < 64 * 16; q += 256) {
        const int ch = q >> 4, rg = q & 15;  // 4 rows per piece
        const int m = m0 + rg * 4;
        if (m >= M) continue;
        uint16_t e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = tile[rg * 4 + j][ch];
        uint16_t* d = dst + (long long)(c0 + ch) * ldd + col0 + m;
        if (m + 3 < M && (((uintptr_t)d) & 7) == 0) {
            uint2 w;
            w.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
            w.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
            *reinterpret_cast<uint2*>(d) = w;
        } else {
            for (int j = 0; j < 4 && m + j < M; ++j) d[j] = e[j];
        }
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ src, int lds_, uint16_t* __restrict__ dst,
                                                        long long ldd, int col0, int M, int C, const int* __restrict__ dyn_col) {
    if (dyn_col) col0 = *dyn_col;
    transpose_body(src, lds_, dst, ldd, col0, M, C, blockIdx.x, blockIdx.y);
}

// Prefill: one launch does the rotary embedding of the q/k heads AND the V -> V^T copy (independent column ranges of the
// same fused qkv rows): workgroups [0, n_tr) take one 64x64 transpose tile each, the rest share the RoPE work grid-stride.
template <int HD>
__global__ __launch_bounds__(256) void qkv_post_llm_kernel(uint16_t* __restrict__ x, int ld, int n_heads, const uint16_t* __restrict__ cosb,
                                                           const uint16_t* __restrict__ sinb, int L, uint16_t* __restrict__ kcache,
                                                           int k_first_head, long long kc_head_stride, int pos0, int v_col, int v_ch,
                                                           uint16_t* __restrict__ vt, long long vt_ld, int n_tr) {
    const int b = blockIdx.x;
    if (b < n_tr) {
        const int tm = (L + 63) / 64;
        transpose_body(x + v_col, ld, vt, vt_ld, pos0, L, v_ch, b % tm, b / tm);
    } else {
        rope_llm_body<HD>(x, ld, 0, n_heads, cosb, sinb, L, kcache, k_first_head, kc_head_stride, pos0, nullptr, b - n_tr, gridDim.x - n_tr);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void qkv_post_vit_kernel(uint16_t* __restrict__ x, int ld, int n_heads2, const float* __restrict__ cosf_,
                                                           const float* __restrict__ sinf_, int S, int v_col, int v_ch,
                                                           uint16_t* __restrict__ vt, long long vt_ld, int n_tr) {
    const int b = blockIdx.x;
    if (b < n_tr) {
        const int tm = (S + 63) / 64;
        transpose_body(x + v_col, ld, vt, vt_ld, 0, S, v_ch, b % tm, b / tm);
    } else {
        rope_vit_body<HD>(x, ld, n_heads2, cosf_, sinf_, S, b - n_tr, gridDim.x - n_tr);
    }
}

// Decode step, one token: mRoPE on the q and k heads of the fused qkv row (in place), append the rotated K heads to
// the K cache and the V heads (transposed) to the V^T cache, all at the device-side position.  One workgroup.
// blockIdx.x = sequence of a decode pool (fo1_pool_qkv_post_bf16): row blockIdx.x of qkv [P, ld], state row blockIdx.x.
// PART (fo1_pool_qkv_post_partials_bf16): the row is first built from the split-K planes of fo1_gemm_bf16_partials — bf16(sum_z part[z] + bias),
// the GEMM epilogue's own rounding — in LDS (at most 4096 columns); only the rotated q heads go back to qkv.
template <int HD, bool PART = false>
__global__ __launch_bounds__(256) void decode_qkv_post_kernel(uint16_t* __restrict__ qkv, long long ld, int n_q, int n_kv, const uint16_t* __restrict__ cosb,
                                                              const uint16_t* __restrict__ sinb, const int* __restrict__ st,
                                                              uint16_t* __restrict__ kcache, long long kc_head_stride,
                                                              uint16_t* __restrict__ vtcache, long long vt_row_stride,
                                                              const float* __restrict__ part, int splits, long long plane,
                                                              const uint16_t* __restrict__ bias) {
    constexpr int HC = HD / 16;
    __shared__ __attribute__((aligned(16))) uint16_t lrow[PART ? 4096 : 8];
    const int tid = threadIdx.x;
    if constexpr (PART) {
        const int N = (n_q + 2 * n_kv) * HD;
        const float* pr = part + (long long)blockIdx.x * N;
        for (int c = tid; c < (N >> 3); c += 256) {
            float4 a0 = *reinterpret_cast<const float4*>(pr + c * 8), a1 = *reinterpret_cast<const float4*>(pr + c * 8 + 4);
            for (int z = 1; z < splits; ++z) {
                const float4 b0 = *reinterpret_cast<const float4*>(pr + z * plane + c * 8), b1 = *reinterpret_cast<const float4*>(pr + z * plane + c * 8 + 4);
                a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
                a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
            }
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if (bias) {
                float bf[8];
                unpack8r(*reinterpret_cast<const uint4*>(bias + c * 8), bf);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += bf[j];
            }
            *reinterpret_cast<uint4*>(lrow + c * 8) = pack8r(v);
        }
        __syncthreads();
    }
    qkv += (long long)blockIdx.x * ld;
    st += blockIdx.x * 8;
    const int pos = st[0], row = st[1];
    const int n_rope = (n_q + n_kv) * HC;
    for (int i = tid; i < n_rope; i += 256) {
        const int c = i % HC, hd = i / HC;
        uint16_t* p = qkv + hd * HD;
        const uint16_t* ps = PART ? lrow + hd * HD : p;
        const int d0 = c * 8;
        float a[8], b[8], ca[8], sa[8], cb[8], sb[8], oa[8], ob[8];
        unpack8r(*reinterpret_cast<const uint4*>(ps + d0), a);
        unpack8r(*reinterpret_cast<const uint4*>(ps + d0 + HD / 2), b);
        unpack8r(*reinterpret_cast<const uint4*>(cosb + (long long)row * HD + d0), ca);
        unpack8r(*reinterpret_cast<const uint4*>(sinb + (long long)row * HD + d0), sa);
        unpack8r(*reinterpret_cast<const uint4*>(cosb + (long long)row * HD + d0 + HD / 2), cb);
        unpack8r(*reinterpret_cast<const uint4*>(sinb + (long long)row * HD + d0 + HD / 2), sb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            oa[j] = rb(a[j] * ca[j]) + rb(-b[j] * sa[j]);
            ob[j] = rb(b[j] * cb[j]) + rb(a[j] * sb[j]);
        }
        const uint4 ua = pack8r(oa), ub = pack8r(ob);
        if (!PART || hd < n_q) {
            *reinterpret_cast<uint4*>(p + d0) = ua;
            *reinterpret_cast<uint4*>(p + d0 + HD / 2) = ub;
        }
        if (hd >= n_q) {
            uint16_t* kc = kcache + (long long)(hd - n_q) * kc_head_stride + (long long)pos * HD;
            *reinterpret_cast<uint4*>(kc + d0) = ua;
            *reinterpret_cast<uint4*>(kc + d0 + HD / 2) = ub;
        }
    }
    const uint16_t* v = (PART ? lrow : qkv) + (n_q + n_kv) * HD;
    for (int i = tid; i < n_kv * HD; i += 256) vtcache[(long long)i * vt_row_stride + pos] = v[i];
}

__global__ void decode_advance_kernel(int* __restrict__ st) {
    if (threadIdx.x == 0) {
        const int pos = st[0] + 1;
        st[0] = pos;
        st[1] = st[1] + 1;
        st[4] = pos; st[5] = pos + 1; st[6] = 0; st[7] = pos + 1;
    }
}

}  // namespace fo1

extern "C" {

int fo1_rope_llm_bf16(void* qkv, int ld, int col0, int n_heads, int head_dim, const void* cos_bf16, const void* sin_bf16, int L,
                      void* kcache, int k_first_head, long long kcache_head_stride, int pos0, const int32_t* dyn_state, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && cos_bf16 && sin_bf16, "rope_llm: NULL operand");
    FO1_CHECK_ARG(head_dim == 128, "rope_llm: head_dim %d not built (128)", head_dim);
    FO1_CHECK_ARG(ld % 8 == 0 && col0 % 8 == 0 && n_heads > 0, "rope_llm: bad layout");
    if (L == 0) return FO1_OK;
    const long long total = (long long)L * n_heads * (head_dim / 16);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    FO1_LAUNCH("rope_llm", (double)L * n_heads * head_dim * 4.0, rope_llm_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
               (uint16_t*)qkv, ld, col0, n_heads, (const uint16_t*)cos_bf16, (const uint16_t*)sin_bf16, L, (uint16_t*)kcache,
               k_first_head, kcache_head_stride, pos0, (const int*)dyn_state);
    return FO1_OK;
}

int fo1_rope_vit_bf16(void* qkv, int ld, int n_heads, int head_dim, const float* cos_f32, const float* sin_f32, int S, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && cos_f32 && sin_f32, "rope_vit: NULL operand");
    FO1_CHECK_ARG(head_dim == 80, "rope_vit: head_dim %d not built (80)", head_dim);
    FO1_CHECK_ARG(ld % 8 == 0 && n_heads > 0, "rope_vit: bad layout");
    if (S == 0) return FO1_OK;
    const long long total = (long long)S * (2 * n_heads) * (head_dim / 16);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    FO1_LAUNCH("rope_vit", (double)S * 2 * n_heads * head_dim * 4.0, rope_vit_kernel<80>, dim3(grid), dim3(256), 0,
               (hipStream_t)stream, (uint16_t*)qkv, ld, 2 * n_heads, cos_f32, sin_f32, S);
    return FO1_OK;
}

int fo1_qkv_post_llm_bf16(void* qkv, int ld, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_bf16, const void* sin_bf16, int L,
                          void* kcache, long long kcache_head_stride, void* vtcache, long long vt_row_stride, int pos0, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && cos_bf16 && sin_bf16 && kcache && vtcache, "qkv_post_llm: NULL operand");
    FO1_CHECK_ARG(head_dim == 128, "qkv_post_llm: head_dim %d not built (128)", head_dim);
    FO1_CHECK_ARG(ld % 8 == 0 && n_q_heads > 0 && n_kv_heads > 0 && ld >= (n_q_heads + 2 * n_kv_heads) * head_dim, "qkv_post_llm: bad layout");
    FO1_CHECK_ARG(pos0 >= 0 && vt_row_stride >= pos0 + L, "qkv_post_llm: V^T cache too narrow");
    if (L == 0) return FO1_OK;
    const int nh = n_q_heads + n_kv_heads, v_ch = n_kv_heads * head_dim;
    const long long total = (long long)L * nh * (head_dim / 16);
    const int n_rope = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const int n_tr = cdiv(L, 64) * (v_ch / 64);
    FO1_LAUNCH("qkv_post_llm", (double)L * (nh + n_kv_heads) * head_dim * 4.0, qkv_post_llm_kernel<128>, dim3(n_tr + n_rope), dim3(256), 0,
               (hipStream_t)stream, (uint16_t*)qkv, ld, nh, (const uint16_t*)cos_bf16, (const uint16_t*)sin_bf16, L, (uint16_t*)kcache,
               n_q_heads, kcache_head_stride, pos0, nh * head_dim, v_ch, (uint16_t*)vtcache, vt_row_stride, n_tr);
    return FO1_OK;
}

int fo1_qkv_post_vit_bf16(void* qkv, int ld, int n_heads, int head_dim, const float* cos_f32, const float* sin_f32, int S, void* vt,
                          long long vt_ld, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && cos_f32 && sin_f32 && vt, "qkv_post_vit: NULL operand");
    FO1_CHECK_ARG(head_dim == 80, "qkv_post_vit: head_dim %d not built (80)", head_dim);
    const int d = n_heads * head_dim;
    FO1_CHECK_ARG(ld % 8 == 0 && n_heads > 0 && ld >= 3 * d && d % 64 == 0 && vt_ld >= S, "qkv_post_vit: bad layout");
    if (S == 0) return FO1_OK;
    const long long total = (long long)S * (2 * n_heads) * (head_dim / 16);
    const int n_rope = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const int n_tr = cdiv(S, 64) * (d / 64);
    FO1_LAUNCH("qkv_post_vit", (double)S * 3 * d * 4.0, qkv_post_vit_kernel<80>, dim3(n_tr + n_rope), dim3(256), 0, (hipStream_t)stream,
               (uint16_t*)qkv, ld, 2 * n_heads, cos_f32, sin_f32, S, 2 * d, d, (uint16_t*)vt, vt_ld, n_tr);
    return FO1_OK;
}

int fo1_transpose_bf16(const void* src, int ld_src, void* dst, long long ld_dst, int col0, const int32_t* dyn_col0, int M, int C,
                       void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(src && dst, "transpose: NULL operand");
    FO1_CHECK_ARG(C > 0 && C % 64 == 0 && ld_src % 8 == 0 && ld_src >= C, "transpose: C=%d must be a multiple of 64", C);
    FO1_CHECK_ARG(col0 >= 0 && ld_dst >= col0 + M, "transpose: destination too narrow");
    if (M == 0) return FO1_OK;
    FO1_LAUNCH("transpose", (double)M * C * 4.0, transpose_kernel, dim3(cdiv(M, 64), C / 64), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, col0, M, C, (const int*)dyn_col0);
    return FO1_OK;
}

int fo1_decode_qkv_post_bf16(void* qkv_row, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table, const void* sin_table,
                             const int32_t* state, void* kcache, long long kcache_head_stride, void* vtcache, long long vt_row_stride,
                             void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv_row && cos_table && sin_table && state && kcache && vtcache, "decode_qkv_post: NULL operand");
    FO1_CHECK_ARG(head_dim == 128, "decode_qkv_post: head_dim %d not built (128)", head_dim);
    FO1_LAUNCH("decode_qkv_post", (double)(n_q_heads + 2 * n_kv_heads) * head_dim * 4.0, decode_qkv_post_kernel<128>, dim3(1), dim3(256), 0,
               (hipStream_t)stream, (uint16_t*)qkv_row, 0LL, n_q_heads, n_kv_heads, (const uint16_t*)cos_table, (const uint16_t*)sin_table,
               (const int*)state, (uint16_t*)kcache, kcache_head_stride, (uint16_t*)vtcache, vt_row_stride, (const float*)nullptr, 0, 0LL,
               (const uint16_t*)nullptr);
    return FO1_OK;
}

// The same for the P sequences of a decode pool (llm.DecodePool): row b of qkv [P, ld] is rotated with table row state[b][1], its
// K heads / V heads land in the caches at row / column state[b][0] (the sequence's own slot).
int fo1_pool_qkv_post_bf16(void* qkv, long long ld, int P, int n_q_heads, int n_kv_heads, int head_dim, const void* cos_table,
                           const void* sin_table, const int32_t* state, void* kcache, long long kcache_head_stride, void* vtcache,
                           long long vt_row_stride, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && cos_table && sin_table && state && kcache && vtcache && P >= 1, "pool_qkv_post: NULL operand");
    FO1_CHECK_ARG(head_dim == 128 && ld % 8 == 0 && ((uintptr_t)qkv & 15) == 0, "pool_qkv_post: head_dim %d / ld %lld not built (128, ld %% 8)", head_dim, ld);
    FO1_LAUNCH("pool_qkv_post", (double)P * (n_q_heads + 2 * n_kv_heads) * head_dim * 4.0, decode_qkv_post_kernel<128>, dim3(P), dim3(256), 0,
               (hipStream_t)stream, (uint16_t*)qkv, ld, n_q_heads, n_kv_heads, (const uint16_t*)cos_table, (const uint16_t*)sin_table,
               (const int*)state, (uint16_t*)kcache, kcache_head_stride, (uint16_t*)vtcache, vt_row_stride, (const float*)nullptr, 0, 0LL,
               (const uint16_t*)nullptr);
    return FO1_OK;
}

// The same fed by the split-K planes of the q/k/v projection (fo1_gemm_bf16_partials): row b = bf16(sum_z part[z][b] + bias); the rotated q
// heads are written to q_out [P, ld] (the decode attention's query rows), K / V^T go straight to the caches.
int fo1_pool_qkv_post_partials_bf16(const float* part, int splits, const void* bias, void* q_out, long long ld, int P, int n_q_heads,
                                    int n_kv_heads, int head_dim, const void* cos_table, const void* sin_table, const int32_t* state, void* kcache,
                                    long long kcache_head_stride, void* vtcache, long long vt_row_stride, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(part && q_out && cos_table && sin_table && state && kcache && vtcache && P >= 1 && splits >= 1, "pool_qkv_post_partials: NULL operand");
    FO1_CHECK_ARG(head_dim == 128 && ld % 8 == 0 && ((uintptr_t)q_out & 15) == 0 && ((uintptr_t)part & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0),
                  "pool_qkv_post_partials: head_dim %d / ld %lld not built (128, ld %% 8, 16-byte aligned operands)", head_dim, ld);
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    FO1_CHECK_ARG(N <= 4096 && ld >= n_q_heads * head_dim, "pool_qkv_post_partials: %d fused columns (<= 4096)", N);
    FO1_LAUNCH("pool_qkv_post_partials", (double)P * N * (4.0 * splits + 2.0), (decode_qkv_post_kernel<128, true>), dim3(P), dim3(256), 0,
               (hipStream_t)stream, (uint16_t*)q_out, ld, n_q_heads, n_kv_heads, (const uint16_t*)cos_table, (const uint16_t*)sin_table,
               (const int*)state, (uint16_t*)kcache, kcache_head_stride, (uint16_t*)vtcache, vt_row_stride, part, splits, (long long)P * N,
               (const uint16_t*)bias);
    return FO1_OK;
}

// Decode bookkeeping on the device so one captured hipGraph serves every step: state = int32[8]
//   [0] cache position, [1] rope-table row, [4..7] the attention work item {q_start, q_end, 0, kv_end}
int fo1_decode_advance(int32_t* state, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(state != nullptr, "decode_advance: NULL state");
    FO1_LAUNCH("decode_advance", 32.0, decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)state);
    return FO1_OK;
}

}  // extern "C"
