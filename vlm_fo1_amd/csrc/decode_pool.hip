// decode_pool.hip — the decode step of a POOL of 64 / 128 sequences on gfx950 (continuous batching: SURVEY 8f-1; reference loop
// being replaced: the 1-token fast path of omchat_qwen2_5_vl.py:143-155 + HF greedy search, positions modeling_qwen2_5_vl.py:1848-1860).
//
// decode_mfma.hip streams the weights once per step for up to 32 sequences (they ride as MFMA columns of a 16-row weight unit, x lives
// in LDS / registers).  Past 32 the x image no longer fits a CU and a 16-row unit would pull the whole x through L2 per KB of weights,
// so the pool step is a real — if skinny — GEMM:  C[P, N] = X[P, K] W[N, K]^T  with P = 64 / 128 sequence slots and the weights
// still read from HBM exactly once per step (6.17 GB / step whatever P is: 4x the sequences of the 32-column step per weight byte).
//
//   pool_gemm_kernel<NSG, MODE>   workgroup = 4 waves = 128 weight rows x all P slots x a K range; wave w owns rows 32 w .. 32 w + 31.
//       W  global -> VGPRs directly, non-temporal, in MFMA-fragment shape: lane (r = lane & 31, h = lane >> 5) reads the 128 contiguous
//          bytes W[row r][kt * 128 + h * 64 .. + 63] of a 128-element K tile as 8 x 16 B; piece j is the A operand of k-substep j
//          (v_mfma_f32_32x32x16_bf16 sums over k, so any k assignment is valid as long as X uses the same one: substep j covers
//          k in {h * 64 + j * 8 + 0..7 : h = 0, 1}).  No LDS round trip for the streamed operand; 3 K tiles (24 KB per wave) in flight.
//       X  (the P activations rows, L2-resident) is staged per K tile through a double-buffered LDS image shared by the 4 waves
//          (row pitch 272 B: conflict-free 16-byte fragment reads), loaded one tile ahead.
//       One barrier per K tile.  No global load sits under a branch (clamped addresses; tail tiles carry no loads).
//       MODE PARTIAL: fp32 partial sums of this workgroup's K range -> part[split][slot][N] (few-row projections q/k/v, o, down are
//          split over K so that every CU streams); PLAIN: bias -> bf16 (+ residual); SWIGLU: 16-row interleaved gate / up rows meet in
//          one lane's accumulator registers.
//   pool_reduce_qkv_kernel        fixed-order sum of the K splits -> bias -> bf16 -> mRoPE (table row state[b][1]) -> rotated q rows out,
//                                 K rows / V^T columns appended to the caches at state[b][0]
//   pool_reduce_res_norm_kernel   fixed-order sum -> bf16 -> + residual -> bf16 = the new hidden row, and Qwen2RMSNorm of it (the next
//                                 projection's input: post_attention_layernorm after o, the next layer's input_layernorm / the final norm
//                                 after down) in the same launch
// Per (slot, feature) the sum order depends on the shape only (K tiles in order inside a split, splits in order): a sequence decodes to
// the same ids in any slot and next to any other sequences.  Against the <= 32-sequence kernels the fp32 order differs (ids may differ
// at near-ties, like any two bf16 executions).
#include "decode_common.h"
#include "ab.h"

namespace fo1 {

typedef __attribute__((ext_vector_type(8))) __bf16 pl_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pl_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int pl_u32x4;

enum { PL_PARTIAL = 0, PL_PLAIN = 1, PL_SWIGLU = 2 };

struct PoolGemmParams {
    const uint16_t* X; long long ldx;      // [P, K]
    const uint16_t* W; long long ldw;      // [N, K]
    int N, K;
    int n_tiles;                           // N / 128
    int kper, splits;                      // K tiles (128 elements) per split
    float* part;                           // PARTIAL: [splits][P][N]
    const uint16_t* bias;                  // PLAIN / SWIGLU
    const uint16_t* res; long long ldr;    // PLAIN
    uint16_t* C; long long ldc;            // PLAIN: [P, N]; SWIGLU: [P, N / 2]
};

constexpr int PL_XP = 272;                 // bytes per staged x row: 256 + 16 (16-byte fragment reads of 16 consecutive rows hit 16 distinct 4-bank groups)

__device__ __forceinline__ uint4 pl_load_nt16(const uint16_t* p) {
    const pl_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pl_u32x4*>(p));
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float pl_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <int NSG, int MODE, bool NT>
__global__ __launch_bounds__(256, 2) void pool_gemm_kernel(const PoolGemmParams p) {
    constexpr int P = NSG * 32;
    constexpr int XBUF = P * PL_XP;
    constexpr int XL = P / 16;                                          // 16-byte x loads per thread and K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];   // [2][P][PL_XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile_n = blockIdx.x % p.n_tiles, split = blockIdx.x / p.n_tiles;   // neighbours share the split's x columns in L2
    const int KT = p.K >> 7;
    const int kt0 = split * p.kper;
    const int kt1 = min(KT, kt0 + p.kper);
    const int nt = kt1 - kt0;                                           // >= 1 (host: splits = ceil(KT / kper))
    const int n0 = tile_n * 128 + wave * 32;
    const uint16_t* const wrow = p.W + (long long)(n0 + r) * p.ldw + h * 64;
    const int xc = tid & 15, xr0 = tid >> 4;                            // x staging: 16-byte chunk of the tile row, first row (rows xr0 + 16 i)
    const uint16_t* const xsrc = p.X + (long long)xr0 * p.ldx + xc * 8;
    unsigned char* const xdst = pl_smem + xr0 * PL_XP + xc * 16;
    const unsigned char* const xfrag = pl_smem + r * PL_XP + h * 128;   // + buf * XBUF + sg * 32 * PL_XP + j * 16

    auto loadW = [&](int kt, uint4 (&w)[8]) __attribute__((always_inline)) {
        const int k = kt < kt1 ? kt : kt1 - 1;                          // clamped: never a load under a branch
        const uint16_t* s = wrow + (long long)k * 128;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = NT ? pl_load_nt16(s + j * 8) : *reinterpret_cast<const uint4*>(s + j * 8);
    };
    auto loadX = [&](int kt, pl_u32x4 (&x)[XL]) __attribute__((always_inline)) {
        const int k = kt < kt1 ? kt : kt1 - 1;
        const uint16_t* s = xsrc + (long long)k * 128;
#pragma unroll
        for (int i = 0; i < XL; ++i) x[i] = *reinterpret_cast<const pl_u32x4*>(s + (long long)(16 * i) * p.ldx);
    };
    auto storeX = [&](int buf, const pl_u32x4 (&x)[XL]) __attribute__((always_inline)) {
        unsigned char* d = xdst + buf * XBUF;
#pragma unroll
        for (int i = 0; i < XL; ++i) *reinterpret_cast<pl_u32x4*>(d + (16 * i) * PL_XP) = x[i];
    };
    pl_f32x16 acc[NSG];
#pragma unroll
    for (int g = 0; g < NSG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
    auto compute = [&](int buf, const uint4 (&w)[8]) __attribute__((always_inline)) {
        const unsigned char* f = xfrag + buf * XBUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint4 b[NSG];
#pragma unroll
            for (int g = 0; g < NSG; ++g) b[g] = *reinterpret_cast<const uint4*>(f + g * (32 * PL_XP) + j * 16);
#pragma unroll
            for (int g = 0; g < NSG; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const pl_bf16x8*>(&w[j]), *reinterpret_cast<const pl_bf16x8*>(&b[g]), acc[g], 0, 0, 0);
        }
    };

    // ---- prologue: x tile 0 staged, W tiles 0 and 1 and x tile 1 in flight ----
    uint4 w0[8], w1[8], w2[8];
    pl_u32x4 xr[XL];      // (an array of HIP's uint4 STRUCTS stays in scratch here: SROA gives up on it; the ext-vector type is promoted)
    loadX(kt0, xr);
    loadW(kt0, w0);
    loadW(kt0 + 1, w1);
    storeX(0, xr);
    loadX(kt0 + 1, xr);
    __syncthreads();

    // One K tile: W two tiles ahead and x two tiles ahead are requested, the tile is multiplied, the next x tile goes to the other LDS
    // buffer (last read one tile ago: every wave has passed the barrier since).
#define PL_STEP(T, WC, WN2)                         \
    do {                                            \
        loadW(kt0 + (T) + 2, WN2);                  \
        compute((T) & 1, WC);                       \
        storeX(((T) + 1) & 1, xr);                  \
        loadX(kt0 + (T) + 2, xr);                   \
        __syncthreads();                            \
    } while (0)
    int t = 0;
    for (; t + 3 <= nt; t += 3) {
        PL_STEP(t, w0, w2);
        PL_STEP(t + 1, w1, w0);
        PL_STEP(t + 2, w2, w1);
    }
#undef PL_STEP
    // tail (nt mod 3 tiles): everything they need is already in flight — no loads here, so the branches cost nothing
    const int rem = nt - t;
    if (rem >= 1) {
        compute(t & 1, w0);
        if (rem == 2) {
            storeX((t + 1) & 1, xr);
            __syncthreads();
            compute((t + 1) & 1, w1);
        }
    }

    // ---- epilogue: acc[g][i] = sum_k W[n0 + 8 (i >> 2) + 4 h + (i & 3)][k] x[32 g + r][k] ----
    if constexpr (MODE == PL_PARTIAL) {
        float* base = p.part + ((long long)split * P) * p.N + n0 + 4 * h;
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            float* row = base + (long long)(g * 32 + r) * p.N;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(row + 8 * q) = float4{acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
        }
    } else if constexpr (MODE == PL_SWIGLU) {
        // rows n0 .. n0 + 15 = gate of features n0 / 2 .. + 15, rows n0 + 16 .. + 31 = their up partners: registers i and i + 8 of one lane
        const uint16_t* const bsrc = p.bias ? p.bias : p.W;
        uint2 bg[2], bu[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bg[q] = *reinterpret_cast<const uint2*>(bsrc + n0 + 8 * q + 4 * h);
            bu[q] = *reinterpret_cast<const uint2*>(bsrc + n0 + 16 + 8 * q + 4 * h);
            if (!p.bias) bg[q] = bu[q] = uint2{0, 0};
        }
        auto h4 = [](const uint2& v, int i) -> float { const uint32_t w = (i >> 1) ? v.y : v.x; return (i & 1) ? bf16_hi(w) : bf16_lo(w); };
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            uint16_t* crow = p.C + (long long)(g * 32 + r) * p.ldc + (n0 >> 1) + 4 * h;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gt = pl_round(acc[g][4 * q + i] + h4(bg[q], i)), up = pl_round(acc[g][8 + 4 * q + i] + h4(bu[q], i));
                    o[i] = pl_round(fo1_silu(gt)) * up;
                }
                *reinterpret_cast<uint2*>(crow + 8 * q) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
    } else {
        const uint16_t* const bsrc = p.bias ? p.bias : p.W;
        uint2 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[q] = *reinterpret_cast<const uint2*>(bsrc + n0 + 8 * q + 4 * h);
            if (!p.bias) bv[q] = uint2{0, 0};
        }
        auto h4 = [](const uint2& v, int i) -> float { const uint32_t w = (i >> 1) ? v.y : v.x; return (i & 1) ? bf16_hi(w) : bf16_lo(w); };
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
            const long long m = g * 32 + r;
            uint16_t* crow = p.C + m * p.ldc + n0 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = pl_round(acc[g][4 * q + i] + h4(bv[q], i));
                if (p.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(p.res + m * p.ldr + n0 + 4 * h + 8 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += h4(rv, i);
                }
                *reinterpret_cast<uint2*>(crow + 8 * q) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
    }
}

// ---- reduce + q/k/v epilogue: one workgroup per slot ------------------------------------------------------------------------------
// part [splits][P][N] with N = (n_q + 2 n_kv) * 128; item = 4 consecutive dims d0 .. d0 + 3 < 64 of a q / k head together with their
// rotary partners d + 64, or 4 consecutive V dims.
struct PoolQkvParams {
    const float* part; int splits, P, N;
    const uint16_t* bias;
    int n_q, n_kv;
    const uint16_t* cos_t; const uint16_t* sin_t;
    const int* state;
    uint16_t* q_out; long long ldq;
    uint16_t* kcache; long long kc_head_stride;
    uint16_t* vtcache; long long vt_row_stride;
};

__device__ __forceinline__ float4 pl_sum_splits(const float* p0, long long split_stride, int splits) {
    float4 s = *reinterpret_cast<const float4*>(p0);
    for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p0 + k * split_stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    return s;
}
__device__ __forceinline__ void pl_unpack4(const uint2& v, float (&f)[4]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
}

__global__ __launch_bounds__(256) void pool_reduce_qkv_kernel(const PoolQkvParams p) {
    const int b = blockIdx.x;
    const int* st = p.state + b * 8;
    const int pos = st[0];
    const long long trow = st[1];
    const float* prow = p.part + (long long)b * p.N;
    const long long ss = (long long)p.P * p.N;
    const int n_rope = (p.n_q + p.n_kv) * 16;           // 16 items of 4 dims per head half
    const int n_items = n_rope + p.n_kv * 32;
    const uint16_t* const bsrc = p.bias ? p.bias : p.cos_t;
    for (int it = threadIdx.x; it < n_items; it += 256) {
        if (it < n_rope) {
            const int head = it >> 4, d0 = (it & 15) * 4;
            const int ra = head * 128 + d0;
            const float4 sa4 = pl_sum_splits(prow + ra, ss, p.splits), sb4 = pl_sum_splits(prow + ra + 64, ss, p.splits);
            float a[4] = {sa4.x, sa4.y, sa4.z, sa4.w}, bb[4] = {sb4.x, sb4.y, sb4.z, sb4.w};
            float ba[4], bbv[4], c1[4], s1[4], c2[4], s2[4];
            pl_unpack4(*reinterpret_cast<const uint2*>(bsrc + ra), ba);
            pl_unpack4(*reinterpret_cast<const uint2*>(bsrc + ra + 64), bbv);
            pl_unpack4(*reinterpret_cast<const uint2*>(p.cos_t + trow * 128 + d0), c1);
            pl_unpack4(*reinterpret_cast<const uint2*>(p.sin_t + trow * 128 + d0), s1);
            pl_unpack4(*reinterpret_cast<const uint2*>(p.cos_t + trow * 128 + d0 + 64), c2);
            pl_unpack4(*reinterpret_cast<const uint2*>(p.sin_t + trow * 128 + d0 + 64), s2);
            float oa[4], ob[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = pl_round(a[i] + (p.bias ? ba[i] : 0.f)), y = pl_round(bb[i] + (p.bias ? bbv[i] : 0.f));   // the bf16 q/k nn.Linear stores
                oa[i] = pl_round(x * c1[i]) + pl_round(-y * s1[i]);                                                        // rotate_half, three bf16 roundings
                ob[i] = pl_round(y * c2[i]) + pl_round(x * s2[i]);
            }
            const uint2 pa = uint2{pack_bf16x2(oa[0], oa[1]), pack_bf16x2(oa[2], oa[3])}, pb = uint2{pack_bf16x2(ob[0], ob[1]), pack_bf16x2(ob[2], ob[3])};
            if (head < p.n_q) {
                uint16_t* q = p.q_out + (long long)b * p.ldq + ra;
                *reinterpret_cast<uint2*>(q) = pa;
                *reinterpret_cast<uint2*>(q + 64) = pb;
            } else {
                uint16_t* kc = p.kcache + (long long)(head - p.n_q) * p.kc_head_stride + (long long)pos * 128 + d0;
                *reinterpret_cast<uint2*>(kc) = pa;
                *reinterpret_cast<uint2*>(kc + 64) = pb;
            }
        } else {
            const int v0 = (it - n_rope) * 4;                         // kv_head * 128 + d
            const int col = (p.n_q + p.n_kv) * 128 + v0;
            const float4 s4 = pl_sum_splits(prow + col, ss, p.splits);
            const float v[4] = {s4.x, s4.y, s4.z, s4.w};
            float bv[4];
            pl_unpack4(*reinterpret_cast<const uint2*>(bsrc + col), bv);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                p.vtcache[(long long)(v0 + i) * p.vt_row_stride + pos] = f32_to_bf16(v[i] + (p.bias ? bv[i] : 0.f));
        }
    }
}

// ---- reduce + residual + RMSNorm: one workgroup per slot, N = 8 * 256 * NV features --------------------------------------------------
struct PoolResNormParams {
    const float* part; int splits, P, N;
    const uint16_t* res; long long ldr;      // [P, N] or null
    uint16_t* x_out; long long ldx;          // [P, N]: bf16(bf16(sum) + residual)
    const uint16_t* norm_w; float eps;       // null: no norm output
    uint16_t* h_out; long long ldh;          // [P, N]: Qwen2RMSNorm(x_out) * norm_w
};

__global__ __launch_bounds__(256) void pool_reduce_res_norm_kernel(const PoolResNormParams p) {
    __shared__ float s_part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* prow = p.part + (long long)b * p.N;
    const long long ss = (long long)p.P * p.N;
    // thread t owns features 4 (t + 256 i) .. + 3, i < N / 1024  (N = 2048: two float4 per split)
    float ssq = 0.f;
    float xv[4][4];      // N <= 4096
    const int nv = p.N >> 10;
    const uint16_t* const rsrc = p.res ? p.res + (long long)b * p.ldr : reinterpret_cast<const uint16_t*>(prow);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nv) {
            const int f = 4 * (tid + 256 * i);
            const float4 s4 = pl_sum_splits(prow + f, ss, p.splits);
            float r4[4];
            pl_unpack4(*reinterpret_cast<const uint2*>(rsrc + f), r4);
            const float s[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = pl_round(s[j]);
                if (p.res) v = pl_round(v + r4[j]);
                xv[i][j] = v;
                ssq = fmaf(v, v, ssq);
            }
            *reinterpret_cast<uint2*>(p.x_out + (long long)b * p.ldx + f) = uint2{pack_bf16x2(xv[i][0], xv[i][1]), pack_bf16x2(xv[i][2], xv[i][3])};
        }
    }
    if (!p.norm_w) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = ssq;
    __syncthreads();
    const float tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float rstd = rsqrtf(tot / (float)p.N + p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nv) {
            const int f = 4 * (tid + 256 * i);
            float w4[4];
            pl_unpack4(*reinterpret_cast<const uint2*>(p.norm_w + f), w4);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = w4[j] * pl_round(xv[i][j] * rstd);      // fp32 variance, bf16(x * rstd), * weight -> bf16 (modeling_qwen2_5_vl.py:126-140)
            *reinterpret_cast<uint2*>(p.h_out + (long long)b * p.ldh + f) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

FO1_AB_VAR g_pool_variant = 0;      // bit 0: W loads non-temporal (bypass L1) instead of L1-allocating (A/B: fo1_pool_gemm_set_variant)

template <int NSG, int MODE, bool NT>
static int launch_pool_gemm_nt(const PoolGemmParams& p, const char* name, hipStream_t st) {
    constexpr int smem = 2 * NSG * 32 * PL_XP;
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)pool_gemm_kernel<NSG, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    FO1_LAUNCH(name, (double)p.N * p.K * 2.0, (pool_gemm_kernel<NSG, MODE, NT>), dim3(p.n_tiles * p.splits), dim3(256), smem, st, p);
    return FO1_OK;
}

template <int NSG, int MODE>
static int launch_pool_gemm(const PoolGemmParams& p, const char* name, hipStream_t st) {
#ifdef FO1_ENABLE_AB
    if (g_pool_variant & 1) return launch_pool_gemm_nt<NSG, MODE, true>(p, name, st);
#endif
    return launch_pool_gemm_nt<NSG, MODE, false>(p, name, st);
}

}  // namespace fo1

extern "C" {

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: test / bench build only
int fo1_pool_gemm_set_variant(int bits) {
    if (bits < 0 || bits > 1) return fo1::set_err(FO1_ERR_ARG, "pool_gemm_set_variant: %d", bits);
    fo1::g_pool_variant = bits;
    return FO1_OK;
}
#endif

// Split policy of the pool GEMM (shape only): N / 128 row tiles; few-row projections are split over K until about one workgroup
// per CU streams, at least two 128-element K tiles per workgroup.  Returns the number of splits; *kper = K tiles per split.
int fo1_pool_gemm_splits(int N, int K, int* kper_out) {
    const int tiles = N / 128, KT = K / 128;
    int splits = 1;
    if (tiles > 0 && tiles < 128) {
        splits = (256 + tiles - 1) / tiles;
        if (splits > KT / 2) splits = KT / 2;
        if (splits < 1) splits = 1;
    }
    const int kper = (KT + splits - 1) / splits;
    if (kper_out) *kper_out = kper;
    return kper > 0 ? (KT + kper - 1) / kper : 1;
}

size_t fo1_pool_gemm_workspace_bytes(int P, int N, int K) {
    int kper;
    const int splits = fo1_pool_gemm_splits(N, K, &kper);
    return splits > 1 ? (size_t)splits * P * N * sizeof(float) : 0;
}

// C[P, N] = epilogue(x[P, K] W[N, K]^T) for the P = 64 / 128 slots of a decode pool; weights streamed once.
//   mode 0 (plain): bias -> bf16 -> + residual -> bf16 (few-row shapes are split over K: fp32 partials in `workspace`, summed in fixed
//           order by the reduce launch, which can also emit Qwen2RMSNorm(C) * norm_weight into norm_out [P, N] — the next projection's input)
//   mode 1 (SwiGLU): W rows interleaved [gate 16 | up 16]; C has N / 2 columns
//   mode 2 (QKV): bias -> bf16 -> mRoPE (table row state[b][1]) -> rotated q rows to C [P, n_q * 128]; K rows / V^T columns appended to the
//           caches at row / column state[b][0]
int fo1_pool_gemm_bf16(const void* x, long long ldx, const void* W, long long ldw, const void* bias, const void* residual, long long ldr,
                       void* C, long long ldc, int P, int N, int K, int mode, const void* norm_weight, float norm_eps, void* norm_out,
                       long long ld_norm, int n_q_heads, int n_kv_heads, const void* cos_table, const void* sin_table, const int32_t* state,
                       void* kcache, long long kcache_head_stride, void* vtcache, long long vt_row_stride, void* workspace,
                       size_t workspace_bytes, void* stream) {
    using namespace fo1;
    hipStream_t st = (hipStream_t)stream;
    FO1_CHECK_ARG(x && W && C, "pool_gemm: NULL operand");
    FO1_CHECK_ARG(P == 64 || P == 128, "pool_gemm: P = %d (a pool has 64 or 128 slots)", P);
    FO1_CHECK_ARG(N > 0 && N % 128 == 0 && K > 0 && K % 128 == 0, "pool_gemm: N = %d and K = %d must be multiples of 128", N, K);
    FO1_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0, "pool_gemm: misaligned operand");
    FO1_CHECK_ARG(mode >= 0 && mode <= 2, "pool_gemm: mode %d", mode);
    FO1_CHECK_ARG(((uintptr_t)bias & 7) == 0 && ((uintptr_t)residual & 7) == 0 && ((uintptr_t)C & 7) == 0 && ldc % 4 == 0 && ldr % 4 == 0,
                  "pool_gemm: epilogue operands must be 8-byte aligned");
    PoolGemmParams g;
    g.X = (const uint16_t*)x; g.ldx = ldx; g.W = (const uint16_t*)W; g.ldw = ldw; g.N = N; g.K = K; g.n_tiles = N / 128;
    g.splits = fo1_pool_gemm_splits(N, K, &g.kper);
    if (mode == 0 && (N % 1024 != 0 || N > 4096) && !norm_weight) {      // the reduce launch's shapes: anything else runs unsplit
        g.splits = 1;
        g.kper = K / 128;
    }
    g.part = (float*)workspace; g.bias = (const uint16_t*)bias; g.res = (const uint16_t*)residual; g.ldr = ldr; g.C = (uint16_t*)C; g.ldc = ldc;
    const bool split = g.splits > 1;
    if (split || mode == 2 || norm_weight)
        FO1_CHECK_ARG(workspace && workspace_bytes >= (size_t)g.splits * P * N * sizeof(float), "pool_gemm: workspace too small (%zu < %zu)",
                      workspace_bytes, (size_t)g.splits * P * N * sizeof(float));
    if (mode == 1) {
        FO1_CHECK_ARG(!split && residual == nullptr && N % 32 == 0, "pool_gemm: SwiGLU shapes are not split over K (N = %d, K = %d)", N, K);
        return P == 64 ? launch_pool_gemm<2, PL_SWIGLU>(g, "pool_gemm_swiglu", st) : launch_pool_gemm<4, PL_SWIGLU>(g, "pool_gemm_swiglu", st);
    }
    if (mode == 0 && !split && !norm_weight)
        return P == 64 ? launch_pool_gemm<2, PL_PLAIN>(g, "pool_gemm", st) : launch_pool_gemm<4, PL_PLAIN>(g, "pool_gemm", st);
    // partial sums + reduce launch
    int rc = P == 64 ? launch_pool_gemm<2, PL_PARTIAL>(g, mode == 2 ? "pool_gemm_qkv" : "pool_gemm_part", st)
                     : launch_pool_gemm<4, PL_PARTIAL>(g, mode == 2 ? "pool_gemm_qkv" : "pool_gemm_part", st);
    if (rc != FO1_OK) return rc;
    if (mode == 2) {
        FO1_CHECK_ARG(n_q_heads > 0 && n_kv_heads > 0 && N == (n_q_heads + 2 * n_kv_heads) * 128, "pool_gemm: QKV mode needs N = (n_q + 2 n_kv) * 128");
        FO1_CHECK_ARG(cos_table && sin_table && state && kcache && vtcache && residual == nullptr, "pool_gemm: QKV mode operands");
        FO1_CHECK_ARG(((uintptr_t)cos_table & 7) == 0 && ((uintptr_t)sin_table & 7) == 0 && ((uintptr_t)kcache & 7) == 0, "pool_gemm: QKV tables must be 8-byte aligned");
        PoolQkvParams q;
        q.part = g.part; q.splits = g.splits; q.P = P; q.N = N; q.bias = g.bias; q.n_q = n_q_heads; q.n_kv = n_kv_heads;
        q.cos_t = (const uint16_t*)cos_table; q.sin_t = (const uint16_t*)sin_table; q.state = (const int*)state; q.q_out = g.C; q.ldq = ldc;
        q.kcache = (uint16_t*)kcache; q.kc_head_stride = kcache_head_stride; q.vtcache = (uint16_t*)vtcache; q.vt_row_stride = vt_row_stride;
        FO1_LAUNCH("pool_reduce_qkv", (double)g.splits * P * N * 4.0, pool_reduce_qkv_kernel, dim3(P), dim3(256), 0, st, q);
        return FO1_OK;
    }
    FO1_CHECK_ARG(bias == nullptr, "pool_gemm: a K-split plain product carries no bias (none of the decode step's has one)");
    FO1_CHECK_ARG(N % 1024 == 0 && N <= 4096, "pool_gemm: the reduce launch handles N = 1024 .. 4096 in steps of 1024 (N = %d)", N);
    FO1_CHECK_ARG(!norm_weight || (norm_out && ((uintptr_t)norm_weight & 7) == 0 && ((uintptr_t)norm_out & 7) == 0 && ld_norm % 4 == 0), "pool_gemm: norm operands");
    PoolResNormParams r;
    r.part = g.part; r.splits = g.splits; r.P = P; r.N = N; r.res = g.res; r.ldr = ldr; r.x_out = g.C; r.ldx = ldc;
    r.norm_w = (const uint16_t*)norm_weight; r.eps = norm_eps; r.h_out = (uint16_t*)norm_out; r.ldh = ld_norm;
    FO1_LAUNCH("pool_reduce_res_norm", (double)g.splits * P * N * 4.0, pool_reduce_res_norm_kernel, dim3(P), dim3(256), 0, st, r);
    return FO1_OK;
}

}  // extern "C"
