// gemv.hip — weight-streaming GEMV for the decode step (M <= 4 rows): the same contract as
// fo1_gemm_bf16 (nn.Linear semantics + fused epilogue incl. the interleaved-SwiGLU form), but HBM-bound:
// every weight byte is read exactly once, 16 B per lane, 16-32 loads in flight per lane; x lives in LDS;
// fp32 accumulate, wave-shuffle reduction, one lane writes.  No MFMA: at M = 1 the matrix cores would run
// at 1/64 utilisation and the LDS round trip is pure overhead (guide: "GEMV / M <= 16: load straight to
// VGPRs, deep unroll").  Algorithmic bytes per launch = N*K*2 (+ x, out).
#include "common.h"

namespace fo1 {

struct GemvParams {
    const uint16_t* X;     // [M, ldx]
    const uint16_t* W;     // [N, ldw]
    const uint16_t* bias;  // [N] or null
    const uint16_t* res;   // [M, ldr] or null
    uint16_t* C;           // [M, ldc]
    int M, N, K, ldx, ldw, ldc, ldr, act;
};

__device__ __forceinline__ float gv_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float gv_act(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 2) return v / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ float gv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dot8(const uint4& w, const uint4& x, float acc) {
    acc = fmaf(bf16_lo(w.x), bf16_lo(x.x), acc); acc = fmaf(bf16_hi(w.x), bf16_hi(x.x), acc);
    acc = fmaf(bf16_lo(w.y), bf16_lo(x.y), acc); acc = fmaf(bf16_hi(w.y), bf16_hi(x.y), acc);
    acc = fmaf(bf16_lo(w.z), bf16_lo(x.z), acc); acc = fmaf(bf16_hi(w.z), bf16_hi(x.z), acc);
    acc = fmaf(bf16_lo(w.w), bf16_lo(x.w), acc); acc = fmaf(bf16_hi(w.w), bf16_hi(x.w), acc);
    return acc;
}

// One wave = NR weight rows (4 plain; 8 for the SwiGLU form: 4 gate rows + their 4 up partners 16 rows further).
template <int MM, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvParams p) {
    constexpr int NR = SWIGLU ? 8 : 4;
    constexpr int U = SWIGLU ? 2 : 4;      // chunks per row in flight
    extern __shared__ __attribute__((aligned(16))) uint16_t sx[];   // [MM][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kch = p.K >> 3;
    for (int i = tid; i < MM * kch; i += 256) {
        const int m = i / kch, c = i - m * kch;
        *reinterpret_cast<uint4*>(&sx[m * p.K + c * 8]) =
            m < p.M ? *reinterpret_cast<const uint4*>(p.X + (long long)m * p.ldx + c * 8) : uint4{0, 0, 0, 0};
    }
    __syncthreads();
    const int unit = blockIdx.x * 4 + wave;
    // rows of this wave
    int rows[NR];
    const int n_feat = SWIGLU ? p.N / 2 : p.N;
    const int f0 = unit * 4;
    if (f0 >= n_feat) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = f0 + j;
        if (f >= n_feat) f = n_feat - 1;          // clamp (result discarded)
        if (SWIGLU) {
            rows[j] = (f >> 4) * 32 + (f & 15);
            rows[4 + j] = rows[j] + 16;
        } else {
            rows[j] = f;
        }
    }
    float acc[NR][MM];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;

    for (int c0 = lane; c0 < kch; c0 += 64 * U) {
        uint4 w[NR][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64;
            const bool ok = c < kch;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                w[r][u] = ok ? *reinterpret_cast<const uint4*>(p.W + (long long)rows[r] * p.ldw + c * 8) : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64;
            if (c < kch) {
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(&sx[m * p.K + c * 8]);
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc[r][m] = dot8(w[r][u], xv, acc[r][m]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[r][m] = gv_wave_sum(acc[r][m]);
    if (lane != 0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + j;
        if (f >= n_feat) break;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m >= p.M) break;
            float v;
            if (SWIGLU) {
                float g = acc[j][m], u = acc[4 + j][m];
                if (p.bias) { g += bf16_to_f32(p.bias[rows[j]]); u += bf16_to_f32(p.bias[rows[4 + j]]); }
                g = gv_round(g);
                u = gv_round(u);
                v = gv_round(g / (1.0f + expf(-g))) * u;
            } else {
                v = acc[j][m];
                if (p.bias) v += bf16_to_f32(p.bias[f]);
                v = gv_round(v);
                if (p.act) v = gv_round(gv_act(v, p.act));
                if (p.res) v += bf16_to_f32(p.res[(long long)m * p.ldr + f]);
            }
            p.C[(long long)m * p.ldc + f] = f32_to_bf16(v);
        }
    }
}

template <int MM>
static int launch_gemv(const GemvParams& p, hipStream_t st) {
    const size_t smem = (size_t)MM * p.K * 2;
    const double bytes = (double)p.N * p.K * 2.0;
    if (p.act == 3) {
        const int units = cdiv(p.N / 2, 4);
        static bool attr = false;
        if (!attr) { FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_kernel<MM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); attr = true; }
        FO1_LAUNCH("gemv_bf16", bytes, (gemv_kernel<MM, true>), dim3(cdiv(units, 4)), dim3(256), smem, st, p);
    } else {
        const int units = cdiv(p.N, 4);
        static bool attr = false;
        if (!attr) { FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_kernel<MM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); attr = true; }
        FO1_LAUNCH("gemv_bf16", bytes, (gemv_kernel<MM, false>), dim3(cdiv(units, 4)), dim3(256), smem, st, p);
    }
    return FO1_OK;
}

// called from gemm_dispatch's front end (gemm.hip) when M <= 4
int gemv_dispatch(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr, void* C, int ldc,
                  int M, int N, int K, int act, hipStream_t st) {
    GemvParams p;
    p.X = (const uint16_t*)A; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual;
    p.C = (uint16_t*)C; p.M = M; p.N = N; p.K = K; p.ldx = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
    if (M == 1) return launch_gemv<1>(p, st);
    if (M == 2) return launch_gemv<2>(p, st);
    return launch_gemv<4>(p, st);
}

}  // namespace fo1
