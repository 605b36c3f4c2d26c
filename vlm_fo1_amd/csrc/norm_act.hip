// norm_act.hip — row-wise normalisations and small elementwise ops (HBM-bound; 16-byte bf16x8
// accesses, one wave per row, wave-shuffle reductions).  Rounding points mirror the reference's
// op-by-op bf16 tensors:
//   Qwen2RMSNorm        modeling_qwen2_5_vl.py:126-140  fp32 variance; x*rsqrt -> bf16; then * weight -> bf16
//   nn.LayerNorm        modeling_davit.py:29-48,357     fp32 statistics, one rounding
//   channel LayerNorm   simple_fpn.py:58-78             (biased variance, eps inside sqrt)
//   SwiGLU              modeling_qwen2_5_vl.py:85-86,636  bf16(silu(g)) * u -> bf16
#include "common.h"

namespace fo1 {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    return u;
}

constexpr int kMaxChunksPerLane = 8;  // D <= 64 lanes * 8 chunks * 8 elements = 4096

// mode 0: RMSNorm (w only); mode 1: LayerNorm (w, b).  NPER = 16-byte chunks per lane (a template parameter: straight-line code),
// LPR = lanes per row: 64, or 32 with two rows per wave when a row is at most 32 chunks (D <= 256: DaViT stage 0, where one row per
// wave left half the lanes idle).  Every load of the row, the weight and the bias is issued up front from a clamped address — the
// first form loaded each chunk inside `if (c < nchunk)` next to its use, i.e. load -> s_waitcnt vmcnt(0) -> use per chunk: NPER
// serialised HBM round trips per row.  Lanes past the row's end contribute exact zeros; the lane-local order (chunk, element) and the
// butterfly are unchanged.  Every multiply-add of the statistics and of the LayerNorm output is an explicit fmaf: under -ffp-contract
// hipcc may fuse either product of (a a) + (b b), and did so differently here and in the fused depthwise-conv + LayerNorm kernel
// (vision_ops.hip), which must agree with this one bit for bit (one element in 200 000 differed by an ulp of its rstd).
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int MODE, int NPER, int LPR>
__global__ __launch_bounds__(256) void rownorm_kernel(const uint16_t* __restrict__ x, int ldx, const uint16_t* __restrict__ w,
                                                      const uint16_t* __restrict__ b, uint16_t* __restrict__ y, int ldy,
                                                      int M, int D, float eps, const int* __restrict__ y_rows) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane & (LPR - 1);
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool row_ok = row < M;           // (no early return: the wave's other row may exist and the shuffles want every lane)
    const int nchunk = D >> 3;
    const uint16_t* xr = x + (size_t)(row_ok ? row : M - 1) * ldx;
    uint4 v[NPER], wv[NPER], bv[NPER];
    bool ok[NPER];
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int c = sub + i * LPR;
        ok[i] = c < nchunk;
        const int cc = ok[i] ? c : 0;
        v[i] = *reinterpret_cast<const uint4*>(xr + cc * 8);
        wv[i] = *reinterpret_cast<const uint4*>(w + cc * 8);
        if (MODE == 1) bv[i] = *reinterpret_cast<const uint4*>(b + cc * 8);
    }
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        float f[8];
        unpack8(v[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = ok[i] ? f[j] : 0.f;
            s += t;
            ss = fmaf(t, t, ss);      // explicit: left to -ffp-contract, hipcc is free to fuse EITHER product of (t0 t0) + (t1 t1)
        }
    }
    float mean = 0.f, rstd;
    if (MODE == 0) {
        ss = row_sum<LPR>(ss);
        rstd = rsqrtf(ss / (float)D + eps);
    } else {
        s = row_sum<LPR>(s);
        mean = s / (float)D;
        // two-pass variance from registers (no catastrophic cancellation)
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = ok[i] ? f[j] - mean : 0.f;
                q = fmaf(d, d, q);
            }
        }
        q = row_sum<LPR>(q);
        rstd = rsqrtf(q / (float)D + eps);
    }
    // y_rows (fo1_layernorm_rows_bf16): row m lands at row y_rows[m] of y — the zero-padded map an implicit-GEMM convolution reads
    uint16_t* yr = y + (size_t)(y_rows ? y_rows[row_ok ? row : M - 1] : row) * ldy;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        float f[8], wf[8], o[8];
        unpack8(v[i], f);
        unpack8(wv[i], wf);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = wf[j] * bf16_to_f32(f32_to_bf16(f[j] * rstd));
        } else {
            float bf[8];
            unpack8(bv[i], bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf((f[j] - mean) * rstd, wf[j], bf[j]);
        }
        if (ok[i] && row_ok) *reinterpret_cast<uint4*>(yr + (sub + i * LPR) * 8) = pack8(o);
    }
}

// Split-K consumer of the decode pool (fo1_gemm_bf16_partials): per row m
//     x_out[m] = bf16( bf16( sum_z part[z][m] (+ bias) ) + residual[m] )        (the rounding points of the GEMM's own epilogue)
//     xn_out[m] = RMSNorm(x_out[m]) * w                                         (Qwen2RMSNorm: bf16(x * rstd) * w, as rownorm_kernel<0>)
// in ONE launch instead of gemm_splitk_reduce + rmsnorm.  One 256-thread workgroup per row, NPER 8-column chunks per thread; the planes
// are summed in z order and the squares in a fixed tree, so a row's result depends on nothing but that row.  x_out may be the residual
// buffer itself (every element is read and then written by the same thread).
template <int NPER>
__global__ __launch_bounds__(256) void splitk_residual_rmsnorm_kernel(const float* __restrict__ part, int splits, long long plane, int N,
                                                                      const uint16_t* __restrict__ bias, const uint16_t* res, int ldr,
                                                                      uint16_t* x_out, int ldx, const uint16_t* __restrict__ w, float eps,
                                                                      uint16_t* __restrict__ xn_out, int ldn) {
    __shared__ float wsum[4];
    const int m = blockIdx.x, tid = threadIdx.x, nchunk = N >> 3;
    float xv[NPER][8];
    uint4 wv[NPER];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int c = tid + i * 256;
        const bool ok = c < nchunk;
        const int cc = ok ? c : 0;
        const float* pr = part + (long long)m * N + cc * 8;
        float4 a0 = *reinterpret_cast<const float4*>(pr), a1 = *reinterpret_cast<const float4*>(pr + 4);
        for (int z = 1; z < splits; ++z) {
            const float4 b0 = *reinterpret_cast<const float4*>(pr + z * plane), b1 = *reinterpret_cast<const float4*>(pr + z * plane + 4);
            a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
            a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
        }
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(res + (size_t)m * ldr + cc * 8), r);
        wv[i] = *reinterpret_cast<const uint4*>(w + cc * 8);
        if (bias) {
            float bf[8];
            unpack8(*reinterpret_cast<const uint4*>(bias + cc * 8), bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bf[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(v[j])) + r[j]));
            xv[i][j] = ok ? t : 0.f;
            ss = fmaf(xv[i][j], xv[i][j], ss);
        }
        if (ok) *reinterpret_cast<uint4*>(x_out + (size_t)m * ldx + c * 8) = pack8(xv[i]);
    }
    ss = row_sum<64>(ss);
    if ((tid & 63) == 0) wsum[tid >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf(((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)N + eps);
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int c = tid + i * 256;
        if (c >= nchunk) continue;
        float wf[8], o[8];
        unpack8(wv[i], wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wf[j] * bf16_to_f32(f32_to_bf16(xv[i][j] * rstd));
        *reinterpret_cast<uint4*>(xn_out + (size_t)m * ldn + c * 8) = pack8(o);
    }
}

// Split-K consumer for the gate/up projection of the decode pool: the planes hold the product against the 16-row interleaved weight
// [gate 16 | up 16 | ...] (ops.interleave_gate_up); out[m, f] = bf16( bf16(silu(bf16(g))) * bf16(u) ) with g, u = sum_z of the planes'
// columns 32 (f / 16) + f % 16 and + 16 — the rounding points of fo1_gemm_bf16's ACT_SWIGLU16 epilogue.  8 features per thread.
__global__ __launch_bounds__(256) void splitk_swiglu_kernel(const float* __restrict__ part, int splits, long long plane, int M, int N,
                                                            uint16_t* __restrict__ out, int ldo) {
    const int F = N >> 1, chunks = F >> 3;
    const long long total = (long long)M * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i - (long long)m * chunks);
        const int f0 = c * 8, col = (f0 >> 4) * 32 + (f0 & 15);
        const float* pr = part + (long long)m * N + col;
        float4 g0 = *reinterpret_cast<const float4*>(pr), g1 = *reinterpret_cast<const float4*>(pr + 4);
        float4 u0 = *reinterpret_cast<const float4*>(pr + 16), u1 = *reinterpret_cast<const float4*>(pr + 20);
        for (int z = 1; z < splits; ++z) {
            const float* pz = pr + z * plane;
            const float4 a0 = *reinterpret_cast<const float4*>(pz), a1 = *reinterpret_cast<const float4*>(pz + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pz + 16), b1 = *reinterpret_cast<const float4*>(pz + 20);
            g0.x += a0.x; g0.y += a0.y; g0.z += a0.z; g0.w += a0.w; g1.x += a1.x; g1.y += a1.y; g1.z += a1.z; g1.w += a1.w;
            u0.x += b0.x; u0.y += b0.y; u0.z += b0.z; u0.w += b0.w; u1.x += b1.x; u1.y += b1.y; u1.z += b1.z; u1.w += b1.w;
        }
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, u[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sg = bf16_to_f32(f32_to_bf16(fo1_silu(bf16_to_f32(f32_to_bf16(g[j])))));
            o[j] = sg * bf16_to_f32(f32_to_bf16(u[j]));
        }
        *reinterpret_cast<uint4*>(out + (size_t)m * ldo + f0) = pack8(o);
    }
}

template <int MODE>
static int launch_rownorm(const char* name, const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int M, int D, float eps,
                          hipStream_t st, const int* y_rows = nullptr) {
    const int nchunk = D >> 3;
    const double bytes = (double)M * D * 4.0;
#define FO1_ROWNORM(NPER, LPR)                                                                                                          \
    FO1_LAUNCH(name, bytes, (rownorm_kernel<MODE, NPER, LPR>), dim3(cdiv(M, 4 * (64 / LPR))), dim3(256), 0, st, (const uint16_t*)x, ldx, \
               (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)y, ldy, M, D, eps, y_rows)
    if (nchunk <= 32) FO1_ROWNORM(1, 32);
    else if (nchunk <= 64) FO1_ROWNORM(1, 64);
    else if (nchunk <= 128) FO1_ROWNORM(2, 64);
    else if (nchunk <= 192) FO1_ROWNORM(3, 64);
    else if (nchunk <= 256) FO1_ROWNORM(4, 64);
    else FO1_ROWNORM(kMaxChunksPerLane, 64);
#undef FO1_ROWNORM
    return FO1_OK;
}

// out[m, f] = bf16( bf16(silu(g[m,f])) * u[m,f] ),  gu = [g | u] with row stride ldgu
__global__ __launch_bounds__(256) void swiglu_kernel(const uint16_t* __restrict__ gu, int ldgu, uint16_t* __restrict__ out,
                                                     int ldo, int M, int F) {
    const int chunks = F >> 3;
    const long long total = (long long)M * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i - (long long)m * chunks);
        float g[8], u[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(gu + (size_t)m * ldgu + c * 8), g);
        unpack8(*reinterpret_cast<const uint4*>(gu + (size_t)m * ldgu + F + c * 8), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s = bf16_to_f32(f32_to_bf16(fo1_silu(g[j])));
            o[j] = s * u[j];
        }
        *reinterpret_cast<uint4*>(out + (size_t)m * ldo + c * 8) = pack8(o);
    }
}

// y = act(x (+ bias)) elementwise over [M, D] rows; act 1 = erf GELU; used by the conv paths
__global__ __launch_bounds__(256) void bias_act_kernel(const uint16_t* __restrict__ x, int ldx, const uint16_t* __restrict__ bias,
                                                       uint16_t* __restrict__ y, int ldy, int M, int D, int act) {
    const int chunks = D >> 3;
    const long long total = (long long)M * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i - (long long)m * chunks);
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (size_t)m * ldx + c * 8), f);
        if (bias) {
            float bf[8];
            unpack8(*reinterpret_cast<const uint4*>(bias + c * 8), bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bf16_to_f32(f32_to_bf16(f[j] + bf[j]));
        }
        if (act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fo1_gelu_erf(f[j]);
        }
        *reinterpret_cast<uint4*>(y + (size_t)m * ldy + c * 8) = pack8(f);
    }
}

// y = bf16(a + b) elementwise over [M, D] rows (DETR-style `with_pos_embed`: tensor + position embedding)
__global__ __launch_bounds__(256) void add_kernel(const uint16_t* __restrict__ a, int lda, const uint16_t* __restrict__ b, int ldb,
                                                  uint16_t* __restrict__ y, int ldy, int M, int D) {
    const int chunks = D >> 3;
    const long long total = (long long)M * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i - (long long)m * chunks);
        float f[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(a + (size_t)m * lda + c * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(b + (size_t)m * ldb + c * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += g[j];
        *reinterpret_cast<uint4*>(y + (size_t)m * ldy + c * 8) = pack8(f);
    }
}

// argmax stage 1: 128 workgroups scan contiguous slices (first index among ties), partial (value, index) to scratch
__global__ __launch_bounds__(256) void argmax_partial_kernel(const uint16_t* __restrict__ x, int n, float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float v = bf16_to_f32(x[i]);
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_v[wave] = best; s_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) { best = s_v[w]; bi = s_i[w]; }
        pv[blockIdx.x] = best;
        pi[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(128) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int np, int* __restrict__ out) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if ((int)threadIdx.x < np) { best = pv[threadIdx.x]; bi = pi[threadIdx.x]; }
    __shared__ float s_v[2];
    __shared__ int s_i[2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_v[1] > best || (s_v[1] == best && s_i[1] < bi)) { best = s_v[1]; bi = s_i[1]; }
        *out = bi;
    }
}

// argmax over a bf16 row (first index among ties, like torch.argmax): one workgroup
__global__ __launch_bounds__(1024) void argmax_kernel(const uint16_t* __restrict__ x, int n, int* __restrict__ out) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = bf16_to_f32(x[i]);
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_v[wave] = best; s_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) { best = s_v[w]; bi = s_i[w]; }
        *out = bi;
    }
}

}  // namespace fo1

namespace fo1 {
__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = uint4{0, 0, 0, 0};
}
// RMSNorm whose output goes straight to the fp8 linear that consumes it (fo1_gemm_fp8): y = Qwen2RMSNorm(x) exactly as rownorm_kernel<0>
// computes it (fp32 variance, bf16(x * rstd) * w -> bf16), then fo1_quantize_rows_e4m3's arithmetic on the bf16 row (scale = absmax / 448,
// q = e4m3(clamp(y / scale))) — bit-identical to the two launches, without the bf16 row ever reaching memory.  One wave per row.
__global__ __launch_bounds__(256) void rmsnorm_quant_e4m3_kernel(const uint16_t* __restrict__ x, int ldx, const uint16_t* __restrict__ w,
                                                                  uint8_t* __restrict__ q, long long ldq, float* __restrict__ scales,
                                                                  int M, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const int nchunk = D >> 3;
    const uint16_t* xr = x + (size_t)row * ldx;
    uint4 v[kMaxChunksPerLane];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunksPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            v[i] = *reinterpret_cast<const uint4*>(xr + c * 8);
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunksPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            float f[8], wf[8], o[8];
            unpack8(v[i], f);
            unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = wf[j] * bf16_to_f32(f32_to_bf16(f[j] * rstd));
            v[i] = pack8(o);                              // the bf16 row rmsnorm would have written
            unpack8(v[i], o);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(o[j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scales[row] = scale;
    uint8_t* qr = q + (long long)row * ldq;
#pragma unroll
    for (int i = 0; i < kMaxChunksPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(f[j] / scale, -448.0f), 448.0f);
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            *reinterpret_cast<uint2*>(qr + c * 8) = uint2{(uint32_t)w0, (uint32_t)w1};
        }
    }
}

}  // namespace fo1

extern "C" {

static int rownorm_check(const void* x, const void* w, const void* y, int M, int D, int ldx, int ldy) {
    using namespace fo1;
    FO1_CHECK_ARG(x && w && y, "norm: NULL operand");
    FO1_CHECK_ARG(M >= 0 && D > 0 && D % 8 == 0 && D <= 4096, "norm: D=%d must be a multiple of 8 and <= 4096", D);
    FO1_CHECK_ARG(ldx >= D && ldy >= D && ldx % 8 == 0 && ldy % 8 == 0, "norm: bad leading dimensions");
    return FO1_OK;
}

int fo1_rmsnorm_bf16(const void* x, int ldx, const void* weight, void* y, int ldy, int M, int D, float eps, void* stream) {
    using namespace fo1;
    int rc = rownorm_check(x, weight, y, M, D, ldx, ldy);
    if (rc) return rc;
    if (M == 0) return FO1_OK;
    return launch_rownorm<0>("rmsnorm", x, ldx, weight, weight, y, ldy, M, D, eps, (hipStream_t)stream);
}

// RMSNorm + row-wise e4m3 quantisation in one launch (the fp8 linears' producer; bit-identical to fo1_rmsnorm_bf16 followed by
// fo1_quantize_rows_e4m3).  q bytes [M, D] (ldq bytes), scales fp32 [M].
int fo1_rmsnorm_quant_e4m3(const void* x, int ldx, const void* weight, int M, int D, float eps, void* q, long long ldq, float* scales,
                           void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && weight && q && scales, "rmsnorm_quant: NULL operand");
    FO1_CHECK_ARG(M >= 0 && D > 0 && D % 8 == 0 && D <= 4096, "rmsnorm_quant: D=%d must be a multiple of 8 and <= 4096", D);
    FO1_CHECK_ARG(ldx >= D && ldx % 8 == 0 && ldq >= D && ldq % 8 == 0 && ((uintptr_t)q & 7) == 0, "rmsnorm_quant: bad leading dimensions");
    if (M == 0) return FO1_OK;
    FO1_LAUNCH("rmsnorm_quant_e4m3", (double)M * D * 3.0, rmsnorm_quant_e4m3_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t*)x, ldx, (const uint16_t*)weight, (uint8_t*)q, ldq, scales, M, D, eps);
    return FO1_OK;
}

int fo1_layernorm_bf16(const void* x, int ldx, const void* weight, const void* bias, void* y, int ldy, int M, int D, float eps,
                       void* stream) {
    using namespace fo1;
    int rc = rownorm_check(x, weight, y, M, D, ldx, ldy);
    if (rc) return rc;
    FO1_CHECK_ARG(bias != nullptr, "layernorm: NULL bias");
    if (M == 0) return FO1_OK;
    return launch_rownorm<1>("layernorm", x, ldx, weight, bias, y, ldy, M, D, eps, (hipStream_t)stream);
}

// LayerNorm whose output row m is written to row y_rows[m] of y (int32 [M]): the producer side of fo1_conv3x3_gemm_bf16 — the normalised map
// goes straight into the zero-padded layout the implicit-GEMM convolution reads (borders: the caller zeroes y once).  Same arithmetic as
// fo1_layernorm_bf16, bit for bit.
int fo1_layernorm_rows_bf16(const void* x, int ldx, const void* weight, const void* bias, void* y, int ldy, const int32_t* y_rows, int M, int D,
                            float eps, void* stream) {
    using namespace fo1;
    int rc = rownorm_check(x, weight, y, M, D, ldx, ldy);
    if (rc) return rc;
    FO1_CHECK_ARG(bias != nullptr && y_rows != nullptr, "layernorm_rows: NULL bias / row map");
    if (M == 0) return FO1_OK;
    return launch_rownorm<1>("layernorm", x, ldx, weight, bias, y, ldy, M, D, eps, (hipStream_t)stream, (const int*)y_rows);
}

int fo1_swiglu_bf16(const void* gate_up, int ldgu, void* out, int ldo, int M, int F, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(gate_up && out, "swiglu: NULL operand");
    FO1_CHECK_ARG(F > 0 && F % 8 == 0 && ldgu >= 2 * F && ldgu % 8 == 0 && ldo >= F && ldo % 8 == 0, "swiglu: bad shape F=%d", F);
    if (M == 0) return FO1_OK;
    const long long total = (long long)M * (F / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    FO1_LAUNCH("swiglu", (double)M * F * 6.0, swiglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t*)gate_up, ldgu, (uint16_t*)out, ldo, M, F);
    return FO1_OK;
}

int fo1_bias_act_bf16(const void* x, int ldx, const void* bias, void* y, int ldy, int M, int D, int act, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && y, "bias_act: NULL operand");
    FO1_CHECK_ARG(D > 0 && D % 8 == 0 && ldx >= D && ldy >= D && ldx % 8 == 0 && ldy % 8 == 0, "bias_act: bad shape D=%d", D);
    FO1_CHECK_ARG(act == 0 || act == 1, "bias_act: act=%d", act);
    if (M == 0) return FO1_OK;
    const long long total = (long long)M * (D / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    FO1_LAUNCH("bias_act", (double)M * D * 4.0, bias_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t*)x, ldx, (const uint16_t*)bias, (uint16_t*)y, ldy, M, D, act);
    return FO1_OK;
}

int fo1_add_bf16(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int M, int D, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(a && b && y, "add: NULL operand");
    FO1_CHECK_ARG(D > 0 && D % 8 == 0 && lda >= D && ldb >= D && ldy >= D && lda % 8 == 0 && ldb % 8 == 0 && ldy % 8 == 0, "add: bad shape D=%d", D);
    if (M == 0) return FO1_OK;
    const long long total = (long long)M * (D / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    FO1_LAUNCH("add", (double)M * D * 6.0, add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, lda, (const uint16_t*)b, ldb,
               (uint16_t*)y, ldy, M, D);
    return FO1_OK;
}

int fo1_argmax_bf16(const void* x, int n, int* out, void* scratch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && out && n > 0, "argmax: bad arguments");
    if (scratch == nullptr || n < 16384) {
        FO1_LAUNCH("argmax", (double)n * 2.0, argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const uint16_t*)x, n, out);
        return FO1_OK;
    }
    // two stages over 128 slices: scratch = 128 floats + 128 ints (1 KiB), caller-owned
    float* pv = (float*)scratch;
    int* pi = (int*)(pv + 128);
    FO1_LAUNCH("argmax", (double)n * 2.0, argmax_partial_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, n, pv, pi);
    FO1_LAUNCH("argmax_final", 1024.0, argmax_final_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, (const float*)pv, (const int*)pi, 128, out);
    return FO1_OK;
}

// Zero-fill as a kernel launch (16-byte stores): what the stage entries use instead of hipMemsetAsync, whose graph nodes faulted on
// replay under stream capture on ROCm 7.2 (profiles/README.md).  p 16-byte aligned, bytes a multiple of 16.
int fo1_zero_bytes(void* p, size_t bytes, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(p != nullptr && ((uintptr_t)p & 15) == 0 && bytes % 16 == 0, "zero_bytes: pointer / size must be 16-byte aligned");
    if (bytes == 0) return FO1_OK;
    const size_t n16 = bytes / 16;
    const int grid = (int)(n16 / 256 + 1 < 4096 ? n16 / 256 + 1 : 4096);
    FO1_LAUNCH("zero_bytes", (double)bytes, zero16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4*)p, n16);
    return FO1_OK;
}

int fo1_splitk_residual_rmsnorm_bf16(const float* part, int splits, int M, int N, const void* bias, const void* residual, int ldr, void* x_out,
                                     int ldx, const void* norm_weight, float eps, void* xn_out, int ldn, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(part && residual && x_out && norm_weight && xn_out, "splitk_residual_rmsnorm: NULL operand");
    FO1_CHECK_ARG(M >= 1 && splits >= 1 && N % 8 == 0 && N >= 8 && N <= 8 * 256 * 4, "splitk_residual_rmsnorm: M=%d splits=%d N=%d (N %% 8, <= 8192)", M, splits, N);
    FO1_CHECK_ARG(ldr % 8 == 0 && ldx % 8 == 0 && ldn % 8 == 0 && ldr >= N && ldx >= N && ldn >= N, "splitk_residual_rmsnorm: row strides");
    FO1_CHECK_ARG(((uintptr_t)part & 15) == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)x_out & 15) == 0 && ((uintptr_t)xn_out & 15) == 0 &&
                      ((uintptr_t)norm_weight & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0), "splitk_residual_rmsnorm: 16-byte alignment");
    const long long plane = (long long)M * N;
    const int nper = cdiv(N >> 3, 256);
#define FO1_SKRN(NP)                                                                                                                        \
    FO1_LAUNCH("splitk_residual_rmsnorm", (double)M * N * (4.0 * splits + 6.0), splitk_residual_rmsnorm_kernel<NP>, dim3(M), dim3(256), 0,  \
               (hipStream_t)stream, part, splits, plane, N, (const uint16_t*)bias, (const uint16_t*)residual, ldr, (uint16_t*)x_out, ldx,   \
               (const uint16_t*)norm_weight, eps, (uint16_t*)xn_out, ldn)
    if (nper == 1) FO1_SKRN(1);
    else if (nper == 2) FO1_SKRN(2);
    else FO1_SKRN(4);
#undef FO1_SKRN
    return FO1_OK;
}

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: a measured no-gain form, test / bench build only
int fo1_splitk_swiglu_bf16(const float* part, int splits, int M, int N, void* out, int ldo, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(part && out && M >= 1 && splits >= 1, "splitk_swiglu: NULL operand");
    FO1_CHECK_ARG(N % 32 == 0 && ldo % 8 == 0 && ldo >= N / 2 && ((uintptr_t)part & 15) == 0 && ((uintptr_t)out & 15) == 0,
                  "splitk_swiglu: N=%d (%% 32), ldo=%d (%% 8, >= N / 2), 16-byte aligned operands", N, ldo);
    const long long total = (long long)M * (N / 16);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    FO1_LAUNCH("splitk_swiglu", (double)M * N * (4.0 * splits + 1.0), splitk_swiglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, part, splits,
               (long long)M * N, M, N, (uint16_t*)out, ldo);
    return FO1_OK;
}
#endif   // FO1_ENABLE_AB

}  // extern "C"
