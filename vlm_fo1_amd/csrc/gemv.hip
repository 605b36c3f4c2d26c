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
    const uint16_t* norm_w;  // optional fused RMSNorm on x (weight [K]); eps below
    float norm_eps;
};

__device__ __forceinline__ float gv_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float gv_act(float v, int act) {
    if (act == 1) return fo1_gelu_erf(v);
    if (act == 2) return fo1_silu(v);
    if (act == 5) return fmaxf(v, 0.0f);
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
// KSPLIT: the 4 waves of a workgroup share ONE unit and split K between them (deep-K projections: 4x the loads in
// flight per weight row, one LDS reduction at the end); otherwise one unit per wave.
template <int MM, bool SWIGLU, bool KSPLIT>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvParams p) {
    constexpr int NR = SWIGLU ? 8 : 4;
    constexpr int U = SWIGLU ? 2 : 4;      // chunks per row in flight
    extern __shared__ __attribute__((aligned(16))) uint16_t sx[];   // [MM][K]
    __shared__ float s_red[4][NR * MM];
    __shared__ float s_ss[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kch = p.K >> 3;
    for (int i = tid; i < MM * kch; i += 256) {
        const int m = i / kch, c = i - m * kch;
        *reinterpret_cast<uint4*>(&sx[m * p.K + c * 8]) =
            m < p.M ? *reinterpret_cast<const uint4*>(p.X + (long long)m * p.ldx + c * 8) : uint4{0, 0, 0, 0};
    }
    __syncthreads();
    if (p.norm_w) {
        // fused Qwen2RMSNorm (modeling_qwen2_5_vl.py:126-140) on the staged rows: fp32 variance, bf16(x*rstd), * weight -> bf16
        for (int m = 0; m < MM; ++m) {
            float ss = 0.f;
            for (int c = tid; c < kch; c += 256) {
                const uint4 v = *reinterpret_cast<const uint4*>(&sx[m * p.K + c * 8]);
                ss = dot8(v, v, ss);
            }
            ss = gv_wave_sum(ss);
            if (lane == 0) s_ss[wave] = ss;
            __syncthreads();
            const float rstd = rsqrtf((s_ss[0] + s_ss[1] + s_ss[2] + s_ss[3]) / (float)p.K + p.norm_eps);
            for (int c = tid; c < kch; c += 256) {
                uint4 v = *reinterpret_cast<const uint4*>(&sx[m * p.K + c * 8]);
                const uint4 w = *reinterpret_cast<const uint4*>(p.norm_w + c * 8);
                uint4 o;
                o.x = pack_bf16x2(bf16_lo(w.x) * gv_round(bf16_lo(v.x) * rstd), bf16_hi(w.x) * gv_round(bf16_hi(v.x) * rstd));
                o.y = pack_bf16x2(bf16_lo(w.y) * gv_round(bf16_lo(v.y) * rstd), bf16_hi(w.y) * gv_round(bf16_hi(v.y) * rstd));
                o.z = pack_bf16x2(bf16_lo(w.z) * gv_round(bf16_lo(v.z) * rstd), bf16_hi(w.z) * gv_round(bf16_hi(v.z) * rstd));
                o.w = pack_bf16x2(bf16_lo(w.w) * gv_round(bf16_lo(v.w) * rstd), bf16_hi(w.w) * gv_round(bf16_hi(v.w) * rstd));
                *reinterpret_cast<uint4*>(&sx[m * p.K + c * 8]) = o;
            }
            __syncthreads();
        }
    }
    const int unit = KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
    // rows of this wave
    int rows[NR];
    const int n_feat = SWIGLU ? p.N / 2 : p.N;
    const int f0 = unit * 4;
    if (f0 >= n_feat) return;   // KSPLIT: uniform per workgroup; otherwise per wave (no barrier follows in that mode)
    // chunk range of this wave
    const int kq = KSPLIT ? ((kch + 3) / 4 + 63) / 64 * 64 : kch;
    const int c_begin = KSPLIT ? wave * kq : 0;
    const int c_end = KSPLIT ? min(kch, c_begin + kq) : kch;
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

    for (int c0 = c_begin + lane; c0 < c_end; c0 += 64 * U) {
        uint4 w[NR][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64;
            const bool ok = c < c_end;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                w[r][u] = ok ? *reinterpret_cast<const uint4*>(p.W + (long long)rows[r] * p.ldw + c * 8) : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64;
            if (c < c_end) {
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
    if (KSPLIT) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int m = 0; m < MM; ++m) s_red[wave][r * MM + m] = acc[r][m];
        }
        __syncthreads();
        if (tid != 0) return;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int m = 0; m < MM; ++m)
                acc[r][m] = (s_red[0][r * MM + m] + s_red[1][r * MM + m]) + (s_red[2][r * MM + m] + s_red[3][r * MM + m]);
    }
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
                v = gv_round(fo1_silu(g)) * u;
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

int g_gemv_profile_shapes = 0;

template <int MM, bool SW, bool KS>
static int launch_gemv2(const GemvParams& p, const char* name, hipStream_t st) {
    const size_t smem = (size_t)MM * p.K * 2;
    const int units = cdiv(SW ? p.N / 2 : p.N, 4);
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_kernel<MM, SW, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    FO1_LAUNCH(name, (double)p.N * p.K * 2.0, (gemv_kernel<MM, SW, KS>), dim3(KS ? units : cdiv(units, 4)), dim3(256), smem, st, p);
    return FO1_OK;
}

template <int MM>
static int launch_gemv(const GemvParams& p, hipStream_t st) {
    char pname[48];
    const char* name = "gemv_bf16";
    if (profile_enabled() && g_gemv_profile_shapes) {
        snprintf(pname, sizeof pname, "gemv %dx%dx%d a%d", p.M, p.N, p.K, p.act);
        name = pname;
    }
    const bool ks = p.K >= 4096;   // measured: 2048x11008 at 2.2 TB/s with one wave per 4 rows (6 serial load rounds)
    if (p.act == 3) return ks ? launch_gemv2<MM, true, true>(p, name, st) : launch_gemv2<MM, true, false>(p, name, st);
    return ks ? launch_gemv2<MM, false, true>(p, name, st) : launch_gemv2<MM, false, false>(p, name, st);
}

// called from gemm_dispatch's front end (gemm.hip) when M <= 4
int gemv_dispatch(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr, void* C, int ldc,
                  int M, int N, int K, int act, hipStream_t st, const void* norm_w, float norm_eps) {
    GemvParams p;
    p.norm_w = (const uint16_t*)norm_w; p.norm_eps = norm_eps;
    p.X = (const uint16_t*)A; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual;
    p.C = (uint16_t*)C; p.M = M; p.N = N; p.K = K; p.ldx = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
    if (M == 1) return launch_gemv<1>(p, st);
    if (M == 2) return launch_gemv<2>(p, st);
    return launch_gemv<4>(p, st);
}

}  // namespace fo1

extern "C" {

// GEMV with the full fo1_gemm_bf16 epilogue set and an optional fused RMSNorm on the input rows (decode step:
// input_layernorm / post_attention_layernorm folded into the q/k/v and gate/up projections).  M <= 4.
int fo1_gemv_bf16(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, int ldr, void* C, int ldc,
                  int M, int N, int K, int act, const void* norm_weight, float norm_eps, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && W && C, "gemv: NULL operand");
    FO1_CHECK_ARG(M >= 1 && M <= 4 && N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "gemv: bad shape M=%d N=%d K=%d", M, N, K);
    FO1_CHECK_ARG((size_t)(M > 2 ? 4 : M) * K * 2 <= 150 * 1024, "gemv: x does not fit LDS (M=%d K=%d)", M, K);
    FO1_CHECK_ARG(((act >= 0 && act <= 3) || act == 5) && (act != 3 || (N % 32 == 0 && residual == nullptr)), "gemv: bad act/N");
    FO1_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)norm_weight & 15) == 0, "gemv: misaligned operand");
    return gemv_dispatch(x, ldx, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, (hipStream_t)stream, norm_weight, norm_eps);
}

}  // extern "C"
