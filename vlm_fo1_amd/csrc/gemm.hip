// gemm.hip — bf16 GEMM with fused epilogues on MFMA (gfx950), the dense block of the
// ViT / DaViT / SimpleFPN / projector / LLM-prefill stages.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]^T )        (W is nn.Linear's [out,in] weight)
//
// Replaces torch's F.linear call sites of the reference hot path
// (modeling_qwen2_5_vl.py:79-81,103-110,151-155,176-177,633-635,731-734; modeling_davit.py:63-65,
//  157-158,235-236; multimodal_projector/builder.py:64-71,103-110), bf16 in, fp32 accumulate.
//
// Tiling for wave64 / CDNA4: 256 threads = 4 waves as 2x2; each wave owns a (BM/2)x(BN/2)
// block of 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, K=32 per instruction).  Operands are
// swapped (a = W rows, b = A rows) so a lane ends up with 4 consecutive output columns of one row:
// bias / residual / store are 8-byte vector accesses.  Two staging paths:
//   reg  : global_load_dwordx4 -> VGPR -> ds_write_b128 into a padded tile (any K % 8 == 0)
//   glds : global_load_lds_dwordx4 straight into a double-buffered, XOR-swizzled (on the SOURCE
//          address) lane-linear LDS image, one barrier per K tile (K % 64 == 0)
// blockIdx -> tile mapping is XCD-aware (8 private L2s): each XCD gets a contiguous run of tiles
// that share a W panel.
//
// Epilogue order mirrors the reference's op-by-op bf16 rounding:
//   y = bf16(acc + bias); y = bf16(act(y)); y = bf16(y + residual)
#include "common.h"
#include "ab.h"

#include <type_traits>

namespace fo1 {


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(8))) int v8i32;
typedef __attribute__((ext_vector_type(4))) unsigned int v4u32;

struct GemmParams {
    const uint16_t* A;
    const uint16_t* W;
    const uint16_t* bias;
    const uint16_t* res;
    uint16_t* C;
    float* C32;
    int M, N, K, lda, ldw, ldc, ldr;
    int act;
    int tiles_m, tiles_n;
    long long sA, sW, sC, sR;  // batch strides (elements), blockIdx.y = batch
    int splits, kper;          // split-K: blockIdx.z = split, kper k-tiles (of 64) per split
    float* part;               // fp32 partials [splits][M][N] when splits > 1
    int stages;                // LDS ring depth: 2 = two-stage kernel, 3/4/6 = counted-vmcnt ring
    int debug;                 // ablation (bench only): 1 skip global loads after tile 0, 2 skip MFMA, 4 skip LDS reads + MFMA
    int coal;                  // 256x256 kernels: epilogue staged through LDS and written as whole 128-byte row segments (16-B stores)
    // fp8 (fo1_gemm_fp8): A and W are OCP e4m3 bytes, C = (A W^T) * scale_m[m] * scale_n[n] before the epilogue
    const float* scale_m;
    const float* scale_n;
    // ring kernels with BN = 128 (fo1_gemm_bf16_wtiled): W is a copy pre-tiled as [N / 128][K / 64][128 rows][64] — a K tile of a column tile is ONE
    // contiguous 16 KB block (the decode pool's weight streams: profiles/r04_hbm_stream_patterns.jsonl)
    int w_tiled = 0;
    // 256 x 256 kernels: tile rows per group of the XCD-grouped tile order (tile_coords_grouped_id).  2 (round 5; 8 before): an XCD's ~32
    // concurrent tiles are then 2 tile rows x 16 tile columns — measured in the 25-image pass with two passes in flight
    // (profiles/r05_gemm_group_m_ab.json): 147.7 images/s at 8, 149.8 at 4, 150.1 at 2; the stand-alone products move by 0-4 %
    int gm = 2;
    // fused q/k/v epilogue (fo1_qkv_proj_rope_bf16, EPI 6 / 7): rotary tables (LLM: bf16 cos / sin [M][128]; ViT: fp32 [M][40]), the K cache
    // (LLM), the V^T destination, head counts
    const void* rope_cos = nullptr;
    const void* rope_sin = nullptr;
    uint16_t* kcache = nullptr;
    long long kc_head_stride = 0;
    uint16_t* vt = nullptr;
    long long vt_ld = 0;
    int pos0 = 0, n_q = 0, n_kv = 0;
    // implicit 3x3 convolution (fo1_conv3x3_gemm_bf16, CONV instantiations of the 256 x 256 kernel): A is a zero-PADDED token-major map
    // [.., Wp, Cin]; a_rows[m] = byte offset of output pixel m's tap (0, 0) in it; K tile kt reads tap (ky, kx) = (kt >> conv_lgc) / 3, % 3
    // at channel block kt & ((1 << conv_lgc) - 1): source = a_rows[m] + ky * conv_row_bytes + kx * conv_cin_bytes + block * 128
    const uint32_t* a_rows = nullptr;
    uint32_t conv_row_bytes = 0, conv_cin_bytes = 0;
    int conv_lgc = 0;
};

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_SWIGLU16 = 3, ACT_RELU = 5 };   // (4 = the split-K partial epilogue of the 256x256 kernels)

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_GELU) return fo1_gelu_erf(v);
    if (act == ACT_SILU) return fo1_silu(v);
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

__device__ __forceinline__ float round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }

__device__ __forceinline__ void tile_coords(const GemmParams& p, int& tm, int& tn) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tm = swz % p.tiles_m;
    tn = swz / p.tiles_m;
}

__device__ __forceinline__ void round2_bf16(float& a, float& b) {
    const uint32_t pk = pack_bf16x2(a, b);
    a = bf16_lo(pk);
    b = bf16_hi(pk);
}

// Vector form of the 16x16-fragment epilogue (N % 4 == 0, 8-byte aligned bias / C / residual rows, bf16 output): every bias piece and every
// residual piece of the wave's block is loaded FIRST, unconditionally, from clamped addresses (out-of-range rows / columns are never
// stored) — the general form below loads each of them inside the bounds branch next to its use, which hipcc turns into load ->
// s_waitcnt vmcnt(0) -> use: FM * FN * 5 serialised cache latencies per wave, most of a small-M product's time.  The activation is
// a template parameter behind one wave-uniform switch.  Same arithmetic and rounding points (bias -> bf16 -> act -> bf16 -> +residual
// -> bf16); values are rounded in pairs (v_cvt_pk_bf16_f32).
template <int FM, int FN, int ACT>
__device__ __forceinline__ void epilogue_vec(const GemmParams& p, f32x4 (&acc)[FN][FM], int m_base, int n_base, int lane, long long offC,
                                             long long offR) {
    const int mi = lane & 15, nq = (lane >> 4) * 4;
    uint2 bv[FN], rv[FM][FN];
    if (p.bias) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int n0 = n_base + fn * 16 + nq;
            bv[fn] = *reinterpret_cast<const uint2*>(p.bias + (n0 < p.N ? n0 : 0));
        }
    }
    if (p.res) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = m_base + fm * 16 + mi;
            const uint16_t* rrow = p.res + offR + (long long)(m < p.M ? m : p.M - 1) * p.ldr;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int n0 = n_base + fn * 16 + nq;
                rv[fm][fn] = *reinterpret_cast<const uint2*>(rrow + (n0 < p.N ? n0 : 0));
            }
        }
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m_base + fm * 16 + mi;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int n0 = n_base + fn * 16 + nq;
            float v[4] = {acc[fn][fm][0], acc[fn][fm][1], acc[fn][fm][2], acc[fn][fm][3]};
            if (p.bias) { v[0] += bf16_lo(bv[fn].x); v[1] += bf16_hi(bv[fn].x); v[2] += bf16_lo(bv[fn].y); v[3] += bf16_hi(bv[fn].y); }
            if constexpr (ACT != ACT_NONE) {
                round2_bf16(v[0], v[1]); round2_bf16(v[2], v[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], ACT);
            }
            if (p.res) {
                round2_bf16(v[0], v[1]); round2_bf16(v[2], v[3]);
                v[0] += bf16_lo(rv[fm][fn].x); v[1] += bf16_hi(rv[fm][fn].x); v[2] += bf16_lo(rv[fm][fn].y); v[3] += bf16_hi(rv[fm][fn].y);
            }
            uint2 ov;
            ov.x = pack_bf16x2(v[0], v[1]);
            ov.y = pack_bf16x2(v[2], v[3]);
            if (m < p.M && n0 < p.N) *reinterpret_cast<uint2*>(p.C + offC + (long long)m * p.ldc + n0) = ov;
        }
    }
}

template <int FM, int FN>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[FN][FM], int m_base, int n_base, int lane,
                                         long long offC, long long offR) {
    const int mi = lane & 15, nq = (lane >> 4) * 4;
    const bool vec_ok = (p.ldc % 4 == 0) && ((offC & 3) == 0) && (p.res == nullptr || ((p.ldr % 4 == 0) && ((offR & 3) == 0)));
    if (vec_ok && p.act != ACT_SWIGLU16 && p.C32 == nullptr && p.N % 4 == 0 && ((uintptr_t)p.C & 7) == 0 &&
        (p.bias == nullptr || ((uintptr_t)p.bias & 7) == 0) && (p.res == nullptr || ((uintptr_t)p.res & 7) == 0)) {
        switch (p.act) {     // wave-uniform
            case ACT_GELU: epilogue_vec<FM, FN, ACT_GELU>(p, acc, m_base, n_base, lane, offC, offR); break;
            case ACT_SILU: epilogue_vec<FM, FN, ACT_SILU>(p, acc, m_base, n_base, lane, offC, offR); break;
            case ACT_RELU: epilogue_vec<FM, FN, ACT_RELU>(p, acc, m_base, n_base, lane, offC, offR); break;
            default: epilogue_vec<FM, FN, ACT_NONE>(p, acc, m_base, n_base, lane, offC, offR); break;
        }
        return;
    }
    if (p.act == ACT_SWIGLU16) {
        // W rows are interleaved in 16-row groups [gate 16 | up 16 | gate 16 | ...]: fragment pair (fn, fn+1) holds
        // gate and up of the same 16 features.  out[m, f] = bf16( bf16(silu(bf16(g))) * bf16(u) ), C has N/2 columns.
        if constexpr (FN >= 2) {
            // bias pieces of the lane's feature columns first (N % 32 == 0: a started [gate 16 | up 16] group is whole), not per element
            // inside the bounds branches
            const bool bvec = p.bias != nullptr && ((uintptr_t)p.bias & 7) == 0;
            uint2 bg[FN / 2], bu[FN / 2];
            if (bvec) {
#pragma unroll
                for (int j = 0; j < FN / 2; ++j) {
                    const int n0 = n_base + j * 32 + nq, nc = n0 < p.N ? n0 : nq;
                    bg[j] = *reinterpret_cast<const uint2*>(p.bias + nc);
                    bu[j] = *reinterpret_cast<const uint2*>(p.bias + nc + 16);
                }
            }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m = m_base + fm * 16 + mi;
                if (m >= p.M) continue;
#pragma unroll
                for (int fn = 0; fn + 1 < FN; fn += 2) {
                    const int n0 = n_base + fn * 16 + nq;       // gate column; up column = n0 + 16
                    if (n0 >= p.N) continue;
                    float o[4];
                    const float gbv[4] = {bf16_lo(bg[fn / 2].x), bf16_hi(bg[fn / 2].x), bf16_lo(bg[fn / 2].y), bf16_hi(bg[fn / 2].y)};
                    const float ubv[4] = {bf16_lo(bu[fn / 2].x), bf16_hi(bu[fn / 2].x), bf16_lo(bu[fn / 2].y), bf16_hi(bu[fn / 2].y)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float g = acc[fn][fm][r], u = acc[fn + 1][fm][r];
                        if (bvec) { g += gbv[r]; u += ubv[r]; }
                        else if (p.bias) { g += bf16_to_f32(p.bias[n0 + r]); u += bf16_to_f32(p.bias[n0 + 16 + r]); }
                        g = round_bf16(g);
                        u = round_bf16(u);
                        o[r] = round_bf16(fo1_silu(g)) * u;
                    }
                    uint2 ov;
                    ov.x = pack_bf16x2(o[0], o[1]);
                    ov.y = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(p.C + offC + (long long)m * p.ldc + ((n_base + fn * 16) >> 1) + nq) = ov;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m_base + fm * 16 + mi;
        if (m >= p.M) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int n0 = n_base + fn * 16 + nq;
            if (n0 >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[fn][fm][r];
            const bool full = (n0 + 3 < p.N);
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (full || n0 + r < p.N) v[r] += bf16_to_f32(p.bias[n0 + r]);
            }
            if (p.C32) {  // fp32 output: no rounding, no activation/residual support needed
                float* o = p.C32 + offC + (long long)m * p.ldc + n0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (full || n0 + r < p.N) o[r] = act_apply(v[r], p.act);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = round_bf16(v[r]);
            if (p.act != ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = round_bf16(act_apply(v[r], p.act));
            }
            uint16_t* o = p.C + offC + (long long)m * p.ldc + n0;
            if (full && vec_ok) {
                if (p.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(p.res + offR + (long long)m * p.ldr + n0);
                    v[0] += bf16_lo(rv.x); v[1] += bf16_hi(rv.x);
                    v[2] += bf16_lo(rv.y); v[3] += bf16_hi(rv.y);
                }
                uint2 ov;
                ov.x = pack_bf16x2(v[0], v[1]);
                ov.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(o) = ov;
            } else {
                for (int r = 0; r < 4; ++r) {
                    if (n0 + r >= p.N) break;
                    float t = v[r];
                    if (p.res) t += bf16_to_f32(p.res[offR + (long long)m * p.ldr + n0 + r]);
                    o[r] = f32_to_bf16(t);
                }
            }
        }
    }
}

// split-K: raw fp32 accumulators of split z -> part[z][m][n] (N % 4 == 0 guaranteed by the dispatcher)
template <int FM, int FN>
__device__ __forceinline__ void store_partial(const GemmParams& p, f32x4 (&acc)[FN][FM], int m_base, int n_base, int lane, int z) {
    const int mi = lane & 15, nq = (lane >> 4) * 4;
    float* base = p.part + (long long)z * p.M * p.N;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m_base + fm * 16 + mi;
        if (m >= p.M) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int n0 = n_base + fn * 16 + nq;
            if (n0 >= p.N) continue;
            *reinterpret_cast<float4*>(base + (long long)m * p.N + n0) =
                float4{acc[fn][fm][0], acc[fn][fm][1], acc[fn][fm][2], acc[fn][fm][3]};
        }
    }
}

// ------------------------------------------------------------------------------------------
// register-staged path
// ------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bt_reg_kernel(const GemmParams p) {
    constexpr int BK = 64, LDK = 72;  // 144-B rows: 16-B aligned, conflict-light for ds_read_b128
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int PA = BM * 8 / 256, PB = BN * 8 / 256;
    __shared__ __attribute__((aligned(16))) uint16_t sA[BM * LDK];
    __shared__ __attribute__((aligned(16))) uint16_t sB[BN * LDK];

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long bz = blockIdx.y;
    const uint16_t* A = p.A + bz * p.sA;
    const uint16_t* W = p.W + bz * p.sW;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[PA], rb[PB];
    const int nk = (p.K + BK - 1) / BK;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = tid + i * 256, row = q >> 3, kc = q & 7;
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            const int k = k0 + kc * 8;
            ra[i] = (k < p.K) ? *reinterpret_cast<const uint4*>(A + (long long)gm * p.lda + k) : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = tid + i * 256, row = q >> 3, kc = q & 7;
            int gn = n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            const int k = k0 + kc * 8;
            rb[i] = (k < p.K) ? *reinterpret_cast<const uint4*>(W + (long long)gn * p.ldw + k) : uint4{0, 0, 0, 0};
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int q = tid + i * 256, row = q >> 3, kc = q & 7;
            *reinterpret_cast<uint4*>(&sA[row * LDK + kc * 8]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int q = tid + i * 256, row = q >> 3, kc = q & 7;
            *reinterpret_cast<uint4*>(&sB[row * LDK + kc * 8]) = rb[i];
        }
    };

    gload(0);
    swrite();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 wf[FN], af[FM];
            const int ko = ks * 32 + (lane >> 4) * 8;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                wf[fn] = *reinterpret_cast<const bf16x8*>(&sB[(wn * WN + fn * 16 + (lane & 15)) * LDK + ko]);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
                af[fm] = *reinterpret_cast<const bf16x8*>(&sA[(wm * WM + fm * 16 + (lane & 15)) * LDK + ko]);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            swrite();
            __syncthreads();
        }
    }
    epilogue<FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz * p.sC, bz * p.sR);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA path (global_load_lds_dwordx4), K % 64 == 0
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bt_glds_kernel(const GemmParams p) {
    constexpr int NWV = WGM * WGN;
    constexpr int BK = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int ROWS = BM + BN;          // rows of 128 B per stage
    constexpr int INST = ROWS / 8;         // 1-KiB wave-instructions per stage
    constexpr int IPW = INST / NWV;        // per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][ROWS][128]

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const long long bz = blockIdx.y;
    const uint16_t* A = p.A + bz * p.sA;
    const uint16_t* W = p.W + bz * p.sW;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane source row pointers for this wave's IPW instructions (k offset added per tile)
    const int lr = lane >> 3, lc = lane & 7;
    const uint16_t* src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int R = (wave * IPW + i) * 8 + lr;  // tile row: [0,BM) = A rows, [BM,BM+BN) = W rows
        // XOR swizzle on the SOURCE chunk with (R >> 1) & 7: two 128-B rows share one 256-B bank row, so
        // 16 consecutive rows land on 16 distinct (half, 16-B slot) positions -> conflict-free ds_read_b128
        const int cs = (lc ^ ((R >> 1) & 7)) * 8;
        if (R < BM) {
            int gm = m0 + R;
            gm = gm < p.M ? gm : p.M - 1;
            src[i] = A + (long long)gm * p.lda + cs;
        } else {
            int gn = n0 + (R - BM);
            gn = gn < p.N ? gn : p.N - 1;
            src[i] = W + (long long)gn * p.ldw + cs;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            char* dst = smem + buf * (ROWS * 128) + (wave * IPW + i) * 1024;  // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    const int nk_all = p.K / BK;
    const int kt0 = blockIdx.z * p.kper;
    const int nk = min(nk_all - kt0, p.kper);   // this split's k-tiles: [kt0, kt0 + nk)
    issue(kt0, 0);
    const int dbg = p.debug;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // carries vmcnt(0): tile kt has landed; everyone is done with the other buffer
        if (kt + 1 < nk && !(dbg & 1)) issue(kt0 + kt + 1, (kt + 1) & 1);
        if (dbg & 4) continue;
        const char* sa = smem + (kt & 1) * (ROWS * 128);
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 wf[FN], af[FM];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int R = wn * WN + fn * 16 + (lane & 15);
                wf[fn] = *reinterpret_cast<const bf16x8*>(sb + R * 128 + ((kc ^ ((R >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int R = wm * WM + fm * 16 + (lane & 15);
                af[fm] = *reinterpret_cast<const bf16x8*>(sa + R * 128 + ((kc ^ ((R >> 1) & 7)) << 4));
            }
            if (dbg & 2) {   // keep the LDS reads alive without MFMA
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) asm volatile("" ::"v"(wf[fn]));
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) asm volatile("" ::"v"(af[fm]));
                continue;
            }
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
        }
    }
    if (p.splits > 1) {
        store_partial<FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, blockIdx.z);
        return;
    }
    epilogue<FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz * p.sC, bz * p.sR);
}

// out = epilogue( sum_z part[z] ) in a fixed order (deterministic); 4 consecutive n per thread
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmParams p) {
    const int n4 = p.N >> 2;
    const long long total = (long long)p.M * n4;
    const long long plane = (long long)p.M * p.N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n0 = (int)(i - (long long)m * n4) * 4;
        float4 a = *reinterpret_cast<const float4*>(p.part + (long long)m * p.N + n0);
        for (int z = 1; z < p.splits; ++z) {
            const float4 b = *reinterpret_cast<const float4*>(p.part + z * plane + (long long)m * p.N + n0);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(p.bias[n0 + r]);
        }
        if (p.C32) {
            float* o = p.C32 + (long long)m * p.ldc + n0;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = act_apply(v[r], p.act);
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = round_bf16(v[r]);
        if (p.act != ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = round_bf16(act_apply(v[r], p.act));
        }
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(p.res[(long long)m * p.ldr + n0 + r]);
        }
        uint16_t* o = p.C + (long long)m * p.ldc + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(v[r]);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA path, 3-stage ring: two K tiles in flight across each barrier (counted vmcnt + raw s_barrier;
// __syncthreads() would drain the DMA queue with vmcnt(0)).  All LDS lives in ONE dynamic array.
// ------------------------------------------------------------------------------------------
// WGM x WGN = the 4 waves' layout over the tile (default 2 x 2).  4 x 1 (round 6): every wave owns 32 rows x ALL BN columns, so a tile of
// 96 columns keeps the interleaved [gate 16 | up 16] SwiGLU pairs inside one wave (tile <128, 96>: 230 workgroups for the decode pool's
// 22016-column gate/up product, one per CU, 5 ring stages = 4 K tiles of weights in flight per CU).
template <int BM, int BN, int NS, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(256) void gemm_bt_ring_kernel(const GemmParams p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int BK = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && ((BM + BN) / 8) % 4 == 0, "wave tile in 16 x 16 fragments; DMA pieces divide over 4 waves");
    constexpr int ROWS = BM + BN;
    constexpr int INST = ROWS / 8;
    constexpr int IPW = INST / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS][ROWS][128]

    int tm, tn;
    tile_coords(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const long long bz = blockIdx.y;
    const uint16_t* A = p.A + bz * p.sA;
    const uint16_t* W = p.W + bz * p.sW;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lr = lane >> 3, lc = lane & 7;
    const int nk_all = p.K / BK;
    const uint16_t* src[IPW];
    int kstep[IPW];                        // elements from one K tile to the next: 64, or a whole 128 x 64 block of a pre-tiled W
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int R = (wave * IPW + i) * 8 + lr;
        const int cs = (lc ^ ((R >> 1) & 7)) * 8;
        kstep[i] = BK;
        if (R < BM) {
            int gm = m0 + R;
            gm = gm < p.M ? gm : p.M - 1;
            src[i] = A + (long long)gm * p.lda + cs;
        } else if (BN == 128 && p.w_tiled) {
            src[i] = W + ((long long)tn * nk_all * BN + (R - BM)) * BK + cs;
            kstep[i] = BN * BK;
        } else {
            int gn = n0 + (R - BM);
            gn = gn < p.N ? gn : p.N - 1;
            src[i] = W + (long long)gn * p.ldw + cs;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            char* dst = smem + buf * (ROWS * 128) + (wave * IPW + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long long)kt * kstep[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    const int kt0 = blockIdx.z * p.kper;
    const int nk = min(nk_all - kt0, p.kper);
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(kt0 + t, t);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's loads of tile kt have landed; tiles kt+1 .. kt+NS-2 (IPW loads each) may still be in flight
        const int ahead = min(nk - 1 - kt, NS - 2);
        if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(IPW) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * IPW) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * IPW) : "memory");
        __builtin_amdgcn_s_barrier();   // every wave's tile-kt loads landed; everyone finished computing tile kt-1
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) {
            int nb = buf + NS - 1;
            if (nb >= NS) nb -= NS;
            issue(kt0 + kt + NS - 1, nb);   // overwrites the buffer tile kt-1 was read from
        }
        const char* sa = smem + buf * (ROWS * 128);
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 wf[FN], af[FM];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int R = wn * WN + fn * 16 + (lane & 15);
                wf[fn] = *reinterpret_cast<const bf16x8*>(sb + R * 128 + ((kc ^ ((R >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int R = wm * WM + fm * 16 + (lane & 15);
                af[fm] = *reinterpret_cast<const bf16x8*>(sa + R * 128 + ((kc ^ ((R >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads of tile kt are done before it arrives at the next barrier
        if (++buf == NS) buf = 0;
    }
    if (p.splits > 1) {
        store_partial<FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, blockIdx.z);
        return;
    }
    epilogue<FM, FN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz * p.sC, bz * p.sR);
}

// ------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, wave tile 128 x 64), v_mfma_f32_32x32x16_bf16, LDS-DMA staging with counted
// vmcnt and a ping-pong ("8-phase") schedule — the large-M path (batched prefill: M = images x rows).
//
// A K tile is consumed in 4 phases, one C quadrant (64 x 32 per wave, 8 MFMAs) each; every phase is
//     [load segment: ds_read the quadrant's operand fragments, issue ONE 16-KiB group of the LDS-DMA prefetch]  s_barrier
//     [MFMA segment: 8 x v_mfma_f32_32x32x16_bf16]                                                              s_barrier
// and waves 4-7 run one barrier behind waves 0-3, so each SIMD (it hosts one wave of either half) always has one wave in
// its MFMA segment while the other reads LDS / issues DMA.  The DMA runs up to two K tiles ahead: a tile's four 16-KiB
// groups {A rows 0-63 | A rows 64-127 | W rows 0-31 | W rows 32-63 of every wave} are re-staged one phase after their last
// read (reads are retired with lgkmcnt(0) before the barrier that ends the load segment, so the lagging half cannot still
// be reading), and the only wait is ONE counted s_waitcnt vmcnt(6) per K tile (three groups stay in flight across it).
// LDS: 2 buffers x 4 groups x 16 KiB = 128 KiB, one workgroup per CU.  The LDS image is lane-linear per DMA instruction
// (8 rows x 128 B); the bank-conflict swizzle (chunk ^ ((row >> 1) & 7)) sits on the SOURCE address and on the read.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FO1_P8_BARRIER()                      \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_s_barrier();         \
        asm volatile("" ::: "memory");        \
    } while (0)

// grouped tile order inside each XCD's contiguous run: p.gm tile rows x consecutive tile columns — the ~32 tiles an XCD runs at once
// share p.gm A panels and 32 / p.gm W panels in its L2 (p.gm = 2 since round 5, see GemmParams)
__device__ __forceinline__ void tile_coords_grouped_id(const GemmParams& p, int bid, int& tm, int& tn) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GM = p.gm;
    const int per_group = GM * p.tiles_n;
    const int gid = swz / per_group, in = swz - gid * per_group;
    const int first = gid * GM;
    const int gsz = min(p.tiles_m - first, GM);
    tm = first + in % gsz;
    tn = in / gsz;
}
__device__ __forceinline__ void tile_coords_grouped(const GemmParams& p, int& tm, int& tn) { tile_coords_grouped_id(p, blockIdx.x, tm, tn); }

template <int ACT>
__device__ __forceinline__ float act_apply_t(float v) {
    if constexpr (ACT == ACT_GELU) return fo1_gelu_erf(v);
    if constexpr (ACT == ACT_SILU) return fo1_silu(v);
    return v;
}

// Epilogue for 32x32 accumulator fragments:
//   acc[mf][nf][r] = C[m_base + mf*32 + (lane & 31)][n_base + nf*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
// EPI: 0..2 = bias -> bf16 -> act EPI -> bf16 -> +residual -> bf16 (the reference's rounding points); 3 = interleaved SwiGLU;
// 4 = raw fp32 partials of this K split.  The variant is a template parameter so that each instantiation's 32 unrolled
// store groups stay small (the activations inlined 128 times blow the unroll budget and push the accumulators to scratch).
template <int EPI, int MF, int NF>
__device__ __forceinline__ void epilogue32(const GemmParams& p, f32x16 (&acc)[MF][NF], int m_base, int n_base, int lane, long long offC,
                                           long long offR, int split) {
    const int mi = lane & 31, hi4 = (lane >> 5) * 4;
    if constexpr (EPI == 4) {
        float* base = p.part + (long long)split * p.M * p.N;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int m = m_base + mf * 32 + mi;
            const bool m_ok = m < p.M;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = n_base + nf * 32 + g * 8 + hi4;
                    if (m_ok && n0 < p.N)   // N % 4 == 0 guaranteed by the dispatcher
                        *reinterpret_cast<float4*>(base + (long long)m * p.N + n0) =
                            float4{acc[mf][nf][g * 4 + 0], acc[mf][nf][g * 4 + 1], acc[mf][nf][g * 4 + 2], acc[mf][nf][g * 4 + 3]};
                }
        }
    } else if constexpr (EPI == ACT_SWIGLU16) {
        // 16-row interleaved [gate 16 | up 16]: one 32-row fragment holds gate (r < 8) and up (r >= 8) of the same 16 features
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int m = m_base + mf * 32 + mi;
            const bool m_ok = m < p.M;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int nb = n_base + nf * 32;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int f0 = g * 8 + hi4;   // feature inside the group; gate row nb + f0 + r, up row nb + 16 + f0 + r
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gt = acc[mf][nf][g * 4 + r], up = acc[mf][nf][8 + g * 4 + r];
                        if (p.bias && nb < p.N) { gt += bf16_to_f32(p.bias[nb + f0 + r]); up += bf16_to_f32(p.bias[nb + 16 + f0 + r]); }
                        gt = round_bf16(gt);
                        up = round_bf16(up);
                        o[r] = round_bf16(fo1_silu(gt)) * up;
                    }
                    uint2 ov;
                    ov.x = pack_bf16x2(o[0], o[1]);
                    ov.y = pack_bf16x2(o[2], o[3]);
                    if (m_ok && nb < p.N) *reinterpret_cast<uint2*>(p.C + offC + (long long)m * p.ldc + (nb >> 1) + f0) = ov;
                }
            }
        }
    } else {
        // the dispatcher only routes here when N % 4 == 0 and C / residual rows are 8-byte aligned
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int m = m_base + mf * 32 + mi;
            const bool m_ok = m < p.M;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = n_base + nf * 32 + g * 8 + hi4;
                    const bool ok = m_ok && n0 < p.N;
                    float v[4] = {acc[mf][nf][g * 4 + 0], acc[mf][nf][g * 4 + 1], acc[mf][nf][g * 4 + 2], acc[mf][nf][g * 4 + 3]};
                    if (p.bias && ok) {
                        const uint2 bv = *reinterpret_cast<const uint2*>(p.bias + n0);
                        v[0] += bf16_lo(bv.x); v[1] += bf16_hi(bv.x); v[2] += bf16_lo(bv.y); v[3] += bf16_hi(bv.y);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = round_bf16(v[r]);
                    if constexpr (EPI != ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = round_bf16(act_apply_t<EPI>(v[r]));
                    }
                    if (p.res && ok) {
                        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + offR + (long long)m * p.ldr + n0);
                        v[0] += bf16_lo(rv.x); v[1] += bf16_hi(rv.x); v[2] += bf16_lo(rv.y); v[3] += bf16_hi(rv.y);
                    }
                    uint2 ov;
                    ov.x = pack_bf16x2(v[0], v[1]);
                    ov.y = pack_bf16x2(v[2], v[3]);
                    if (ok) *reinterpret_cast<uint2*>(p.C + offC + (long long)m * p.ldc + n0) = ov;
                }
        }
    }
}

#ifdef FO1_ENABLE_AB
// fo1_gemm_set_debug bit 5 (32): workgroup timeline.  Waves 0 and 7 (one of each staggered half) write the 100 MHz s_memrealtime at kernel
// entry, first MFMA, end of the K loop and end of the epilogue, plus their HW_ID / XCC_ID (and s_memtime at both ends of the K loop), to the buffer of fo1_gemm_set_stamp_buffer:
// [workgroup][2][8] u64 (slots 6, 7: inside the coalesced epilogue — conversions staged in LDS, last store issued).  scripts/gemm_timeline.py turns that into turnover / prologue / K loop / epilogue per workgroup and per CU.
__device__ __forceinline__ void gemm_stamp(const GemmParams& p, int wave, int lane, int slot) {
    if ((p.debug & 32) && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.part) + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
        o[slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) {
            o[4] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            o[5] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
        }
        // the shader-clock counter at both ends of the K loop, low 32 bits in the upper halves of the two id words: cycles / time = the
        // clock the loop actually ran at (DVFS)
        if (slot == 1) o[4] |= (unsigned long long)(unsigned)__builtin_amdgcn_s_memtime() << 32;
        if (slot == 2) o[5] |= (unsigned long long)(unsigned)__builtin_amdgcn_s_memtime() << 32;
    }
}
#define FO1_GEMM_STAMP(slot) gemm_stamp(p, wave, lane, slot)
#else
#define FO1_GEMM_STAMP(slot)
#endif

// Coalesced epilogue of the 256 x 256 kernels: the wave's 128 x 64 output block goes through its own 16 KiB of the (now idle)
// LDS image — bias / activation applied in registers, rows written as 8-byte pieces into a 128-B-per-row image whose 16-B slots
// are XOR-swizzled with the row — and is read back row-wise, 16 B per lane, 8 lanes per row: every global store instruction
// writes 8 whole 128-byte row segments (the fragment-shaped epilogue stores 8 B per lane to 32 different rows per instruction).
// The residual is added in the row-wise phase from 16-B loads.  Same arithmetic and rounding points as epilogue32.
//
// Round 3: NO LOAD SITS INSIDE A PER-ELEMENT BRANCH.  The first form loaded the bias under `if (p.bias && n0 < N)` next to its use
// (32 times per lane, 64 two-byte loads in the SwiGLU form) and the residual row under `if (m < M && n < N)` in each of the 16
// write-out iterations: hipcc cannot hoist a load out of a branch, so every one of them was load -> s_waitcnt vmcnt(0) -> use, a
// chain of 16-64 cache / HBM latencies per tile (residual products: ~16 us of fixed cost per round of tiles against ~9 us without).
// Now the residual rows (16 x 16 B per lane) and the lane's 8 bias pieces are loaded FIRST, unconditionally, from clamped addresses
// (the values of out-of-range rows / columns are never stored), under one wave-uniform branch each; the conversions run underneath
// their latency.  Values are rounded in pairs (one v_cvt_pk_bf16_f32 per two elements; with no activation the rounding IS the
// pack).  Bit-identical output (tests/test_ops_gpu.py: coalesced == fragment-shaped epilogue32, bit for bit).
// Tried on top and dropped (profiles/r03_gemm_epilogue_row_pipelined_ab_no_gain.json): the write-out of each 32-row block issued between the
// conversions of the following blocks (stores spread over the epilogue) — within +-1 % without a residual, 6 % slower with one.
template <int EPI, bool HAS_BIAS, bool HAS_RES>
__device__ __forceinline__ void epilogue32_coalesced_b(const GemmParams& p, f32x16 (&acc)[4][2], char* region, int m_base, int n_base, int lane,
                                                       long long offC, long long offR) {
    const int mi = lane & 31, hi = lane >> 5, hi4 = hi * 4;
    if constexpr (EPI == ACT_SWIGLU16) {
        // 32 output columns per wave: 64-B rows, four 16-B slots swizzled with (row & 3)
        // bias of the lane's features: gate rows nb + f0 .. + 3 and up rows nb + 16 + f0 .. + 3 for the four (nf, g) — the same for every mf
        uint2 bg[4], bu[4];
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nb = n_base + (q >> 1) * 32, f0 = (q & 1) * 8 + hi4;
                const int nc = nb < p.N ? nb + f0 : f0;     // N % 32 == 0 for the interleaved layout: a started 32-row group is whole
                bg[q] = *reinterpret_cast<const uint2*>(p.bias + nc);
                bu[q] = *reinterpret_cast<const uint2*>(p.bias + nc + 16);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nf = q >> 1, g = q & 1;
            float gb[4] = {0.f, 0.f, 0.f, 0.f}, ub[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS) {
                gb[0] = bf16_lo(bg[q].x); gb[1] = bf16_hi(bg[q].x); gb[2] = bf16_lo(bg[q].y); gb[3] = bf16_hi(bg[q].y);
                ub[0] = bf16_lo(bu[q].x); ub[1] = bf16_hi(bu[q].x); ub[2] = bf16_lo(bu[q].y); ub[3] = bf16_hi(bu[q].y);
            }
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int r = mf * 32 + mi;
                float gt[4], up[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { gt[t] = acc[mf][nf][g * 4 + t]; up[t] = acc[mf][nf][8 + g * 4 + t]; }
                if constexpr (HAS_BIAS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { gt[t] += gb[t]; up[t] += ub[t]; }
                }
                round2_bf16(gt[0], gt[1]); round2_bf16(gt[2], gt[3]);
                round2_bf16(up[0], up[1]); round2_bf16(up[2], up[3]);
                float sv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) sv[t] = fo1_silu(gt[t]);
                round2_bf16(sv[0], sv[1]); round2_bf16(sv[2], sv[3]);
                uint2 ov;
                ov.x = pack_bf16x2(sv[0] * up[0], sv[1] * up[1]);
                ov.y = pack_bf16x2(sv[2] * up[2], sv[3] * up[3]);
                *reinterpret_cast<uint2*>(region + r * 64 + ((q ^ (r & 3)) << 4) + hi * 8) = ov;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n_out = (n_base >> 1) + (lane & 3) * 8, No = p.N >> 1;
        uint4 lv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 16 + (lane >> 2);
            lv[it] = *reinterpret_cast<const uint4*>(region + r * 64 + (((lane & 3) ^ (r & 3)) << 4));
        }
        const bool col_ok = n_out < No && !(p.debug & 8);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m_base + it * 16 + (lane >> 2);
            if (m < p.M && col_ok) {
                uint4* dst = reinterpret_cast<uint4*>(p.C + offC + (long long)m * p.ldc + n_out);
                if (p.coal == 2) __builtin_nontemporal_store(*reinterpret_cast<const v4u32*>(&lv[it]), reinterpret_cast<v4u32*>(dst));
                else *dst = lv[it];
            }
        }
    } else {
        // the lane's 8 bias pieces (the same for every 32-row block) first; the residual rows of the write-out phase — all 16 in flight
        // at once — are issued half way through the conversions, when 64 of the 128 accumulator registers are free for them (issued
        // in front, hipcc spills ~100 registers, each reload a vmcnt(0)); their latency runs under the second half.
        const int n = n_base + (lane & 7) * 8;
        uint2 bv[8];
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n0 = n_base + (q >> 2) * 32 + (q & 3) * 8 + hi4;
                bv[q] = *reinterpret_cast<const uint2*>(p.bias + (n0 < p.N ? n0 : 0));
            }
        }
        uint4 rv[16];
        auto load_res = [&]() {
            if (HAS_RES && p.res) {
                const uint16_t* rb = p.res + offR + (n < p.N ? n : 0);
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int m = m_base + it * 8 + (lane >> 3);
                    rv[it] = *reinterpret_cast<const uint4*>(rb + (long long)(m < p.M ? m : p.M - 1) * p.ldr);
                }
            }
        };
        // (the sched_barriers pin the phases: left free, hipcc's scheduler spreads the 16 row loads and the write-out's address
        // arithmetic through the conversions and spills ~100 registers — each reload a vmcnt(0))
        // column group outermost (its 4 bias values unpacked once, live only here), the four 32-row blocks inside
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nf = q >> 2, g = q & 3;
            if (q == 4) {
                __builtin_amdgcn_sched_barrier(0);
                load_res();
                __builtin_amdgcn_sched_barrier(0);
            }
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS) { b[0] = bf16_lo(bv[q].x); b[1] = bf16_hi(bv[q].x); b[2] = bf16_lo(bv[q].y); b[3] = bf16_hi(bv[q].y); }
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int r = mf * 32 + mi;
                float v[4] = {acc[mf][nf][g * 4 + 0], acc[mf][nf][g * 4 + 1], acc[mf][nf][g * 4 + 2], acc[mf][nf][g * 4 + 3]};
                if constexpr (HAS_BIAS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] += b[t];
                }
                if constexpr (EPI != ACT_NONE) {
                    round2_bf16(v[0], v[1]); round2_bf16(v[2], v[3]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = act_apply_t<EPI>(v[t]);
                }
                uint2 ov;       // the pack rounds: bf16(bf16(x)) == bf16(x), so no separate rounding step without an activation
                ov.x = pack_bf16x2(v[0], v[1]);
                ov.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(region + r * 128 + ((q ^ (r & 7)) << 4) + hi * 8) = ov;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FO1_ENABLE_AB
        gemm_stamp(p, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane, 6);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // write-out in two halves of 8 rows, each in three straight sweeps (8 LDS reads in flight, the residual adds under ONE uniform
        // branch, the stores): interleaved per row they were 16 x (ds_read -> wait -> branch -> store)
        const bool col_ok = n < p.N && !(p.debug & 8);
        int lrow = lane >> 3;       // opaque copy: shared with the residual loads' row numbers, hipcc keeps all 16 live (and spills them)
        asm volatile("" : "+v"(lrow));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            __builtin_amdgcn_sched_barrier(0);
            uint4 lv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (hf * 8 + i) * 8 + lrow;
                lv[i] = *reinterpret_cast<const uint4*>(region + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
            }
            if (HAS_RES && p.res) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint4 x = rv[hf * 8 + i];
                    uint4& v = lv[i];
                    v.x = pack_bf16x2(bf16_lo(v.x) + bf16_lo(x.x), bf16_hi(v.x) + bf16_hi(x.x));
                    v.y = pack_bf16x2(bf16_lo(v.y) + bf16_lo(x.y), bf16_hi(v.y) + bf16_hi(x.y));
                    v.z = pack_bf16x2(bf16_lo(v.z) + bf16_lo(x.z), bf16_hi(v.z) + bf16_hi(x.z));
                    v.w = pack_bf16x2(bf16_lo(v.w) + bf16_lo(x.w), bf16_hi(v.w) + bf16_hi(x.w));
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m_base + (hf * 8 + i) * 8 + lrow;
                if (m < p.M && col_ok) {
                    uint4* dst = reinterpret_cast<uint4*>(p.C + offC + (long long)m * p.ldc + n);
                    if (p.coal == 2) __builtin_nontemporal_store(*reinterpret_cast<const v4u32*>(&lv[i]), reinterpret_cast<v4u32*>(dst));
                    else *dst = lv[i];
                }
            }
        }
    }
}

// ---- fused q/k/v epilogue of the 256 x 256 kernel (round 5, VERDICT r4 #1a): bias -> bf16 (the projection's own rounding), then what
// fo1_qkv_post_llm_bf16 / fo1_qkv_post_vit_bf16 did in a second launch over the whole [M, 3 d] activation — rotary embedding of the q / k heads,
// K rows appended to the cache, V written transposed — while the tile still sits in LDS.  Same arithmetic and rounding points as those kernels
// (rope.hip): bit-identical results (tests/test_ops_gpu.py::test_qkv_proj_rope_*).
//   MODE 0, LLM (modeling_qwen2_5_vl.py:643-685,162-169): columns [q heads | k heads | v heads], head dim 128 = two waves of a tile row:
//     rotate-half partner = the same row / slot of the NEIGHBOUR wave's staged block (wave ^ 1); cos / sin bf16 [M][128] per packed row;
//     three roundings bf16(bf16(x cos) + bf16(+-y sin)); q -> C, k -> kcache[kv head][pos0 + m][128], v -> vt[(kv head * 128 + d)][pos0 + m].
//   MODE 1, ViT (:219-230): HEAD-MAJOR columns — one 256-wide tile per head = [q 80 | k 80 | v 80 | 16 pad] (the weight rows are laid out
//     so at load: a head's rotate-half pairs d, d + 40 never straddle two workgroups); fp32 cos / sin [M][40], fp32 math, ONE rounding;
//     q, k -> C (the attention reads them with head stride 256), v -> vt[(head * 80 + d)][pos0 + m].
// The waves of a tile meet at ONE extra workgroup barrier between staging and write-out (a partner slot belongs to another wave).
template <int MODE>
__device__ __forceinline__ void epilogue32_qkv(const GemmParams& p, f32x16 (&acc)[4][2], char* smem, int wave, int m_base, int n_base, int lane) {
    char* region = smem + wave * 16384;
    const int mi = lane & 31, hi = lane >> 5, hi4 = hi * 4;
    // staged block: 128 rows x 128 B, the 16-B slot of column chunk q of row r is q ^ (r & 7) ^ ((r >> 3) & 7): conflict-free for the 16-B row
    // reads of the rotation (8 consecutive rows x 8 chunks per instruction) AND for the 2-byte column reads of the V transpose (8 rows that are
    // 8 apart per instruction)
    auto slot = [](int q, int r) { return (q ^ (r & 7) ^ ((r >> 3) & 7)) << 4; };
    {
        uint2 bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[q] = uint2{0u, 0u};
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) bv[q] = *reinterpret_cast<const uint2*>(p.bias + n_base + (q >> 2) * 32 + (q & 3) * 8 + hi4);      // N % 256 == 0: in range
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nf = q >> 2, g = q & 3;
            const float b[4] = {bf16_lo(bv[q].x), bf16_hi(bv[q].x), bf16_lo(bv[q].y), bf16_hi(bv[q].y)};
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int r = mf * 32 + mi;
                uint2 ov;
                ov.x = pack_bf16x2(acc[mf][nf][g * 4 + 0] + b[0], acc[mf][nf][g * 4 + 1] + b[1]);
                ov.y = pack_bf16x2(acc[mf][nf][g * 4 + 2] + b[2], acc[mf][nf][g * 4 + 3] + b[3]);
                *reinterpret_cast<uint2*>(region + r * 128 + slot(q, r) + hi * 8) = ov;
            }
        }
    }
    const int lrow = lane >> 3, c = lane & 7;
    auto unpack8 = [](const uint4& u, float (&f)[8]) {
        f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
        f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
    };
    // V columns [cg0, cg1) x 8 of this wave's block, transposed: an instruction covers 8 columns x 64 rows — lane = (column, 8-row group), the 8
    // lanes of a column write 128 contiguous bytes of its V^T row.  vcol0 = the V^T row of the block's column 0 (may lie before the buffer when
    // cg0 > 0: only columns >= cg0 * 8 are touched).
    auto store_vt = [&](int cg0, int cg1, uint16_t* vcol0) {
        const int dl = lane >> 3, rg = lane & 7;
        for (int cg = cg0; cg < cg1; ++cg) {
            uint16_t* vrow = vcol0 + (long long)(cg * 8 + dl) * p.vt_ld;
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                uint32_t w[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int r0 = rh * 64 + rg * 8 + 2 * j2;
                    const uint32_t lo = *reinterpret_cast<const uint16_t*>(region + r0 * 128 + ((cg ^ (2 * j2) ^ rg) << 4) + dl * 2);
                    const uint32_t up = *reinterpret_cast<const uint16_t*>(region + (r0 + 1) * 128 + ((cg ^ (2 * j2 + 1) ^ rg) << 4) + dl * 2);
                    w[j2] = lo | (up << 16);
                }
                const int rr = rh * 64 + rg * 8, mg = m_base + rr;
                if (mg + 8 <= p.M) {
                    *reinterpret_cast<uint4*>(vrow + rr) = uint4{w[0], w[1], w[2], w[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (mg + j < p.M) vrow[rr + j] = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
                }
            }
        }
    };
    if constexpr (MODE == 0) {
        const int Nq = p.n_q * 128, Nk = p.n_kv * 128;
        const bool is_v = n_base >= Nq + Nk;                           // wave-uniform
        const bool is_k = !is_v && n_base >= Nq, first = (n_base & 64) == 0;
        const int d0 = (n_base & 64) + c * 8;
        // every table piece of the wave's 16 row groups is requested BEFORE the workgroup barrier (the accumulators are dead: 128 registers):
        // their latency runs under the wait for the tile's other waves
        uint4 cv[16], sv[16];
        if (!is_v) {
            const uint16_t* cosb = reinterpret_cast<const uint16_t*>(p.rope_cos) + d0;
            const uint16_t* sinb = reinterpret_cast<const uint16_t*>(p.rope_sin) + d0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m_base + i * 8 + lrow;
                const long long mr = (long long)(m < p.M ? m : p.M - 1) * 128;
                cv[i] = *reinterpret_cast<const uint4*>(cosb + mr);
                sv[i] = *reinterpret_cast<const uint4*>(sinb + mr);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (is_v) {
            const int kvh = (n_base - Nq - Nk) >> 7;
            store_vt(0, 8, p.vt + (long long)(kvh * 128 + (n_base & 64)) * p.vt_ld + p.pos0 + m_base);
            return;
        }
        const char* preg = smem + (wave ^ 1) * 16384;                  // the head's other half: same rows, same slots
        uint16_t* dst = is_k ? p.kcache + (long long)((n_base - Nq) >> 7) * p.kc_head_stride + (long long)p.pos0 * 128 + d0
                             : p.C + n_base + c * 8;
        const long long dld = is_k ? 128 : p.ldc;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            uint4 xv[8], yv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (hf * 8 + i) * 8 + lrow;
                xv[i] = *reinterpret_cast<const uint4*>(region + r * 128 + slot(c, r));
                yv[i] = *reinterpret_cast<const uint4*>(preg + r * 128 + slot(c, r));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x[8], y[8], cs[8], sn[8];
                unpack8(xv[i], x); unpack8(yv[i], y); unpack8(cv[hf * 8 + i], cs); unpack8(sv[hf * 8 + i], sn);
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a0 = x[2 * j] * cs[2 * j], a1 = x[2 * j + 1] * cs[2 * j + 1];
                    float b0 = (first ? -y[2 * j] : y[2 * j]) * sn[2 * j], b1 = (first ? -y[2 * j + 1] : y[2 * j + 1]) * sn[2 * j + 1];
                    round2_bf16(a0, a1);
                    round2_bf16(b0, b1);
                    o[j] = pack_bf16x2(a0 + b0, a1 + b1);
                }
                const int m = m_base + (hf * 8 + i) * 8 + lrow;
                if (m < p.M) *reinterpret_cast<uint4*>(dst + (long long)m * dld) = uint4{o[0], o[1], o[2], o[3]};
            }
        }
    } else {
        const int wn = wave & 3, wm = wave >> 2, head = n_base >> 8;
        // Rotation work items (round 6): (row r of the tile's 256, 8-angle piece j of the 40 rotary angles).  A lane rotates the q pair (columns 8 j,
        // 40 + 8 j) AND the k pair (80 + 8 j, 120 + 8 j) of its row with ONE 64-byte piece of the fp32 tables: 80 KB of table reads per tile where the
        // column-wise split (a lane = 8 columns of one row, partner columns re-read, every wave its own copy of the tables) read 384 KB — the
        // rotation was bound by those loads (4.6 us of a 9.4 us epilogue, profiles/r06_qkv_epilogue_timeline.json).  1 280 items over the
        // workgroup's 8 waves; same fmaf per element as rope.hip's rope_vit_body.  The tables are requested before the barrier.
        // The four waves that also store V^T afterwards (wn >= 2) take one item per lane, the other four take four: 1 024 + 256 items.
        constexpr int NIT = 4;
        const bool vwave = wn >= 2;
        const int nit = vwave ? 1 : NIT;                                 // wave-uniform
        const int tid = vwave ? 1024 + (wm * 2 + wn - 2) * 64 + lane : (wm * 2 + wn) * 64 + lane;      // first item; the next ones 256 further
        const int m0 = m_base - wm * 128;
        uint16_t* dstC = p.C + (n_base - wn * 64);                       // tile column 0 of this head
        float4 tb[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it >= nit) break;
            const int item = it * 256 + tid;
            const int r = item / 5, j = item - r * 5;
            const int m = m0 + r;
            const long long mr = (long long)(m < p.M ? m : p.M - 1) * 40 + j * 8;
            const float* cp = reinterpret_cast<const float*>(p.rope_cos) + mr;
            const float* sp = reinterpret_cast<const float*>(p.rope_sin) + mr;
            tb[it][0] = *reinterpret_cast<const float4*>(cp); tb[it][1] = *reinterpret_cast<const float4*>(cp + 4);
            tb[it][2] = *reinterpret_cast<const float4*>(sp); tb[it][3] = *reinterpret_cast<const float4*>(sp + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        FO1_GEMM_STAMP(6);        // (timeline: tile staged, every wave past the barrier)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it >= nit) break;
            {
                const int item = it * 256 + tid;
                const int r = item / 5, j = item - r * 5;
                const int lr = r & 127;
                const char* rows = smem + (r >> 7) * 4 * 16384 + lr * 128;
                auto piece = [&](int tcol) { return *reinterpret_cast<const uint4*>(rows + (tcol >> 6) * 16384 + slot((tcol & 63) >> 3, lr)); };
                const int tcol[4] = {8 * j, 40 + 8 * j, 80 + 8 * j, 120 + 8 * j};
                const uint4 q1 = piece(tcol[0]), q2 = piece(tcol[1]), k1 = piece(tcol[2]), k2 = piece(tcol[3]);
                const float cs[8] = {tb[it][0].x, tb[it][0].y, tb[it][0].z, tb[it][0].w, tb[it][1].x, tb[it][1].y, tb[it][1].z, tb[it][1].w};
                const float sn[8] = {tb[it][2].x, tb[it][2].y, tb[it][2].z, tb[it][2].w, tb[it][3].x, tb[it][3].y, tb[it][3].z, tb[it][3].w};
                auto rotate = [&](const uint4& a, const uint4& b, uint4& oa, uint4& ob) {      // (a, b) = (first half, second half) of a head's rotary pair
                    float x[8], y[8], u[8], v[8];
                    unpack8(a, x); unpack8(b, y);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        u[e] = __builtin_fmaf(x[e], cs[e], -(y[e] * sn[e]));       // first half:  x cos - y sin
                        v[e] = __builtin_fmaf(y[e], cs[e], x[e] * sn[e]);          // second half: y cos + x sin
                    }
                    oa = uint4{pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3]), pack_bf16x2(u[4], u[5]), pack_bf16x2(u[6], u[7])};
                    ob = uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                };
                uint4 o[4];
                rotate(q1, q2, o[0], o[1]);
                rotate(k1, k2, o[2], o[3]);
                const int m = m0 + r;
                if (m < p.M) {
                    uint16_t* drow = dstC + (long long)m * p.ldc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<uint4*>(drow + tcol[e]) = o[e];
                }
            }
        }
        FO1_GEMM_STAMP(7);        // (timeline: rotation done — wave 0; wave 7 holds V and pad columns only)
        if (wn >= 2) {      // the head's V columns: tile columns 160..239 = block columns 32..63 of wave 2, 0..47 of wave 3
            uint16_t* vcol0 = p.vt + (long long)(head * 80 + wn * 64 - 160) * p.vt_ld + p.pos0 + m_base;
            if (wn == 2) store_vt(4, 8, vcol0);
            else store_vt(0, 6, vcol0);
        }
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue32_coalesced(const GemmParams& p, f32x16 (&acc)[4][2], char* region, int m_base, int n_base, int lane,
                                                     long long offC, long long offR) {
    // bias presence is a template parameter behind a wave-uniform branch; the residual stays a run-time uniform branch around its 16
    // loads (as a template parameter the straight-line code gave hipcc's scheduler room to spill ~100 registers)
    constexpr bool RES = EPI != ACT_SWIGLU16;      // (the interleaved SwiGLU product has no residual operand)
    if (p.bias) epilogue32_coalesced_b<EPI, true, RES>(p, acc, region, m_base, n_base, lane, offC, offR);
    else epilogue32_coalesced_b<EPI, false, RES>(p, acc, region, m_base, n_base, lane, offC, offR);
}

#ifdef FO1_ENABLE_AB      // four-phase schedule: superseded by gemm_bt_p4_kernel (+3..10 %), kept for A/B only
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bt_p8_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int GROUP = 16384, BUFSZ = 4 * GROUP;       // A0 | A1 | B0 | B1
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUFSZ]

    int tm, tn;
    tile_coords_grouped(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wave >= 4;                              // waves 4-7 run one barrier behind
    const long long bz = blockIdx.y;
    const uint16_t* A = p.A + bz * p.sA;
    const uint16_t* W = p.W + bz * p.sW;

    // ---- DMA source pointers: group g, instruction i (local rows 16*wave + 8*i + (lane >> 3)), chunk swizzled on the source ----
    const uint16_t* src[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = wave * 16 + i * 8 + (lane >> 3);
            const int cs = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;
            if (g < 2) {   // A half g: local row = wm'*64 + j  ->  tile row wm'*128 + g*64 + j
                int gm = m0 + (lr >> 6) * 128 + g * 64 + (lr & 63);
                gm = gm < p.M ? gm : p.M - 1;
                src[g][i] = A + (long long)gm * p.lda + cs;
            } else {       // W half g-2: local row = wn'*32 + j  ->  tile row wn'*64 + (g-2)*32 + j
                int gn = n0 + (lr >> 5) * 64 + (g - 2) * 32 + (lr & 31);
                gn = gn < p.N ? gn : p.N - 1;
                src[g][i] = W + (long long)gn * p.ldw + cs;
            }
        }
    const int nk_all = p.K / BK;
    const int kt0 = blockIdx.z * p.kper;
    const int nk = min(nk_all - kt0, p.kper);
    auto stage = [&](int g, int kt, int buf) {   // one 16-KiB group of k-tile kt0+kt into buffer buf
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = smem + buf * BUFSZ + g * GROUP + (wave * 2 + i) * 1024;   // wave-uniform
            const uint16_t* s = (g == 0 ? src[0][i] : g == 1 ? src[1][i] : g == 2 ? src[2][i] : src[3][i]) + (long long)(kt0 + kt) * BK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a group) ----
    const int hi = lane >> 5;
    int a_off[2], a_sw[2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int lr = wm * 64 + fm * 32 + (lane & 31);
        a_off[fm] = lr * 128;
        a_sw[fm] = (lr >> 1) & 7;
    }
    const int b_lr = wn * 32 + (lane & 31);
    const int b_off = b_lr * 128, b_sw = (b_lr >> 1) & 7;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 areg[2][4], breg[4];

    // ---- prologue: tile 0 complete, tile 1's A0 / B1 / A1 in flight (its B0 is issued in phase 0 of tile 0) ----
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    if (nk > 1) {
        stage(0, 1, 1); stage(3, 1, 1); stage(1, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FO1_P8_BARRIER();
    if (late) FO1_P8_BARRIER();

    auto tile = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        const char* base = smem + BUF * BUFSZ;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        auto loadA = [&](int h) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
                    areg[fm][ks] = *reinterpret_cast<const bf16x8*>(base + h * GROUP + a_off[fm] + (((ks * 2 + hi) ^ a_sw[fm]) << 4));
        };
        auto loadB = [&](int h) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                breg[ks] = *reinterpret_cast<const bf16x8*>(base + (2 + h) * GROUP + b_off + (((ks * 2 + hi) ^ b_sw) << 4));
        };
        auto end_load = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads are retired before it arrives
            __builtin_amdgcn_sched_barrier(0);
            FO1_P8_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto end_mfma = [&](bool last) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(last && late)) FO1_P8_BARRIER();   // the lagging half's final barrier has no partner: drop it
            __builtin_amdgcn_sched_barrier(0);
        };
#define FO1_P8_MFMA(AH, BH)                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) _Pragma("unroll") for (int fm = 0; fm < 2; ++fm)                    \
        acc[(AH) * 2 + fm][BH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[ks], areg[fm][ks], acc[(AH) * 2 + fm][BH], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
        // phase 0: quadrant (a0, b0)
        loadB(0);
        loadA(0);
        if (more1) stage(2, t + 1, BUF ^ 1);
        end_load();
        FO1_P8_MFMA(0, 0)
        end_mfma(false);
        // phase 1: quadrant (a0, b1)
        loadB(1);
        if (more2) stage(0, t + 2, BUF);
        end_load();
        FO1_P8_MFMA(0, 1)
        end_mfma(false);
        // phase 2: quadrant (a1, b1)
        loadA(1);
        if (more2) stage(3, t + 2, BUF);
        end_load();
        FO1_P8_MFMA(1, 1)
        end_mfma(false);
        // phase 3: quadrant (a1, b0); the counted wait that retires tile t+1
        loadB(0);
        if (more2) {
            stage(1, t + 2, BUF);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (more1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        end_load();
        FO1_P8_MFMA(1, 0)
        end_mfma(!more1);
#undef FO1_P8_MFMA
    };
    for (int t = 0; t < nk; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) tile(std::integral_constant<int, 1>{}, t + 1);
    }
    if constexpr (EPI != 4) {
        if (p.coal) {   // every wave's LDS reads of the main loop are retired (the last load segment ended at a barrier this wave has passed)
            epilogue32_coalesced<EPI>(p, acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, lane, bz * p.sC, bz * p.sR);
            return;
        }
    }
    epilogue32<EPI, 4, 2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, bz * p.sC, bz * p.sR, blockIdx.z);
}

#endif   // FO1_ENABLE_AB (gemm_bt_p8_kernel)

FO1_AB_VAR g_gemm_variant = 0;  // 0 auto, 1 reg, 2 glds two-stage, 3/4/6 glds ring of that depth
FO1_AB_VAR g_gemm_tile = 0;     // 0 auto, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 128x256 (8 waves), 5 = 256x256 ping-pong
FO1_AB_VAR g_gemm_splitk = 0;   // 0 auto, n >= 1 forced
static int g_gemm_profile_shapes = 0;   // profile-row naming (fo1_gemm_profile_shapes, include/fo1_ab.h)
FO1_AB_VAR g_gemm_debug = 0;
#ifdef FO1_ENABLE_AB
static void* g_gemm_stamp_buf = nullptr;    // fo1_gemm_set_stamp_buffer (debug bit 5)
#endif
FO1_AB_VAR g_gemm_gemv = 1;      // route M <= 4 to the weight-streaming GEMV (gemv.hip)
FO1_AB_VAR g_gemm_group_m = 0;   // A/B (fo1_gemm_set_group_m): tile rows per group of the 256 x 256 kernels' tile order; 0 = the product rule

extern int g_gemv_profile_shapes;
int gemv_dispatch(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr, void* C, int ldc,
                  int M, int N, int K, int act, hipStream_t st, const void* norm_w, float norm_eps);

// 128 x 256 tile, 8 waves (2 x 4): halves the L2->LDS traffic of the 64 x 128 tile for wide-N GEMMs
template <int BM, int BN>
static int launch_gemm_wide(GemmParams& p, int batch, hipStream_t st, bool reduce = true) {
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(p.N, BN);
    const dim3 grid(p.tiles_m * p.tiles_n, batch, p.splits);
    const double flops = 2.0 * p.M * (double)p.N * p.K * batch;
    char pname[48];
    const char* name = BM == 128 ? "gemm_bt_glds<128,256>" : "gemm_bt_glds<256,256>";
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm %dx%dx%d t%dx%d s%d", p.M, p.N, p.K, BM, BN, p.splits);
        name = pname;
    }
    constexpr int smem = 2 * (BM + BN) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_glds_kernel<BM, BN, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    FO1_LAUNCH(name, flops, (gemm_bt_glds_kernel<BM, BN, 2, 4>), grid, dim3(512), smem, st, p);
    if (p.splits > 1 && reduce) {
        const long long total = (long long)p.M * (p.N / 4);
        const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        FO1_LAUNCH("gemm_splitk_reduce", (double)p.M * p.N * 4.0 * p.splits, gemm_splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
    }
    return FO1_OK;
}

// ------------------------------------------------------------------------------------------
// Same 256 x 256 x 64 tile, same LDS image and DMA groups as gemm_bt_p8_kernel, but TWO fat phases per K tile instead of four:
//     phase X: quadrants (a0,b0) (a0,b1): ds_read A0 (8) + B0 + B1 (8, kept in registers for phase Y)  | 16 MFMAs
//     phase Y: quadrants (a1,b0) (a1,b1): ds_read A1 (8)                                               | 16 MFMAs
// and the LDS-DMA prefetch is issued INSIDE the MFMA segments (one group after every fourth MFMA: a DMA piece costs ~60 cycles
// between MFMAs vs 100-185 in a segment full of ds_reads), so a load segment is ds_reads only and stays well under the 512 cycles
// of its partner's MFMA segment.  Half the barriers (4 per K tile), 24 instead of 28 fragment reads.
//   MFMA-X(t) stages B1, A1 of tile t+1 -> other buffer;  MFMA-Y(t) stages A0, B0 of tile t+2 -> this buffer
//   (each slot's last ds_read — by either wave half — lies at least one barrier earlier)
//   waits: end of load-Y(t): vmcnt(2)  -> A0, B0, B1 of tile t+1 landed (its A1 pair may still fly)
//          end of load-X(t): vmcnt(4)  -> A1 of tile t landed (A0, B0 of tile t+1, issued in MFMA-Y(t-1), may still fly)
//   and each wait sits one barrier before the first read it guards for EITHER half (the lagging half waits one segment later
//   than the leading one and still a barrier ahead of the leading half's read).
// ------------------------------------------------------------------------------------------
// FP8: the same byte geometry (a K tile = 128 B per row = 128 e4m3 elements), v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales
// (E8M0 127) — twice the K per MFMA at twice the rate: 16 MFMAs of 64 cycles per K tile where bf16 issues 32 of 32 cycles, so the DMA
// schedule and every wait count carry over.  A lane's 32 operand bytes are two 16-B slots of the same swizzled LDS image; A and B
// fragments are read the same way, so the operands' common k order inside a lane does not matter.
// ABL (A/B build only, scripts/gemm_loop_ablation.py): main-loop ablations for timing — 1 = no LDS-DMA issue inside the loop, 2 = no fragment
// reads inside the loop, 4 = no barriers inside the loop, 8 = no vmcnt waits inside the loop, 16 = every DMA piece from K tile 0 (L2-hot), 32 = half of a wave's DMA pieces issued in its load-Y segment
// (results valid), 64 = only the A half of the DMA pieces is issued (what a kernel that fetched W another way would leave on the LDS-DMA path).  The results of an ablated launch are garbage by construction.
template <int EPI, bool FP8 = false, int ABL = 0, bool CONV = false>
__global__ __launch_bounds__(512, 2) void gemm_bt_p4_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, ES = FP8 ? 1 : 2, BK = 128 / ES;
    using frag_t = std::conditional_t<FP8, v8i32, bf16x8>;
    constexpr int NKS = FP8 ? 2 : 4;                    // MFMA k-steps per K tile
    constexpr int GROUP = 16384, BUFSZ = 4 * GROUP;       // A0 | A1 | B0 | B1
    // LY (A/B only, ABL & 32): four of a wave's eight DMA pieces per K tile ride in load-Y (8 fragment reads, half a segment of slack) and two
    // in each MFMA segment instead of four.  Measured equal to the product placement on every pass shape (profiles/r04_gemm_loop_ablation_*.json):
    // what the DMA costs the K loop (22-28 %) does not depend on which wave issues a piece or when.
    constexpr bool LY = !FP8 && (ABL & 32) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUFSZ]

    int tm, tn;
    tile_coords_grouped(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wave >= 4;
    FO1_GEMM_STAMP(0);
    const long long bz = blockIdx.y;
    const char* A = reinterpret_cast<const char*>(p.A) + bz * p.sA * ES;
    const char* W = reinterpret_cast<const char*>(p.W) + bz * p.sW * ES;

    // per-lane source of each DMA piece: a 32-bit byte offset from a wave-uniform base.  bf16: the base is this workgroup's first A / W row
    // (+ the K tile), so the offset spans at most 255 rows; the piece is issued in the saddr form (SGPR base pair + one offset VGPR) from
    // inline asm — hipcc selects the 64-bit-VGPR form for __builtin_amdgcn_global_load_lds, whose address pair it rebuilds with a
    // v_lshl_add_u64 per piece IN an operand register the MFMAs before it still read (measured, profiles/r04_gemm_loop_ablation*.json:
    // the DMA issue, not its data, cost 22-28 % of the K loop).  FP8: offsets from the matrix bases (fo1_gemm_fp8 bounds them to 4 GB).
    uint32_t src[4][2];
    const char* At = A + ((FP8 || CONV) ? 0ll : (long long)m0 * p.lda * ES);
    const char* Wt = W + (FP8 ? 0ll : (long long)n0 * p.ldw * ES);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = wave * 16 + i * 8 + (lane >> 3);
            const int cs = ((lane & 7) ^ ((lr >> 1) & 7)) * 16;      // bytes
            if (g < 2) {
                int gm = m0 + (lr >> 6) * 128 + g * 64 + (lr & 63);
                gm = gm < p.M ? gm : p.M - 1;
                if constexpr (CONV) src[g][i] = p.a_rows[gm] + cs;       // implicit convolution: the row's tap-(0, 0) address in the padded map
                else src[g][i] = (uint32_t)(FP8 ? gm : gm - m0) * (uint32_t)(p.lda * ES) + cs;
            } else {
                int gn = n0 + (lr >> 5) * 64 + (g - 2) * 32 + (lr & 31);
                gn = gn < p.N ? gn : p.N - 1;
                src[g][i] = (uint32_t)(FP8 ? gn : gn - n0) * (uint32_t)(p.ldw * ES) + cs;
            }
        }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nk_all = p.K / BK;
    const int kt0 = blockIdx.z * p.kper;
    const int nk = (p.debug & 16) ? 1 : min(nk_all - kt0, p.kper);      // (ablation: one K tile = prologue + epilogue only)
    auto piece = [&](int g, int i, int kt, int buf) __attribute__((always_inline)) {
        {
            const uint32_t so = g == 0 ? src[0][i] : g == 1 ? src[1][i] : g == 2 ? src[2][i] : src[3][i];
            const char* ub;
            if (CONV && g < 2) {        // K tile -> (tap row, tap column, channel block): scalar arithmetic, the lane offsets stay loop-invariant
                const int kk = kt0 + kt, tap = kk >> p.conv_lgc, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
                ub = At + (long long)ky * p.conv_row_bytes + (long long)kx * p.conv_cin_bytes + (long long)(kk & ((1 << p.conv_lgc) - 1)) * 128;
            } else {
                ub = (g < 2 ? At : Wt) + (long long)((ABL & 16) ? 0 : kt0 + kt) * 128;
            }
            const uint32_t dst = lds0 + buf * BUFSZ + g * GROUP + (wave * 2 + i) * 1024;     // M0 = the piece's LDS base
            // (M0 is on the clobber list — ADVICE r4: the compiler must know the statement rewrites it; hipcc accepts the reserved register with a
            // -Winline-asm warning, silenced for this statement.  No compiler-generated M0 use exists in this kernel: its only LDS-DMA is this one.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(so), "s"(ub), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
        }
    };
    auto stage = [&](int g, int kt, int buf) __attribute__((always_inline)) {
        piece(g, 0, kt, buf);
        piece(g, 1, kt, buf);
    };
    // one piece pinned between two MFMA pairs of a segment (the compiler otherwise bunches the segment's four pieces behind its first MFMA)
    auto pinned = [&](bool on, int g, int i, int kt, int buf) __attribute__((always_inline)) {
        if ((ABL & 64) && g >= 2) return;       // ablation: the W half of the DMA traffic is not issued
        if (on) {
            __builtin_amdgcn_sched_barrier(0);
            piece(g, i, kt, buf);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int hi = lane >> 5;
    int a_off[2], a_sw[2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int lr = wm * 64 + fm * 32 + (lane & 31);
        a_off[fm] = lr * 128;
        a_sw[fm] = (lr >> 1) & 7;
    }
    const int b_lr = wn * 32 + (lane & 31);
    const int b_off = b_lr * 128, b_sw = (b_lr >> 1) & 7;
    const int a_base8 = a_off[0] + (((hi * 2) ^ a_sw[0]) << 4), b_base8 = b_off + (((hi * 2) ^ b_sw) << 4);   // FP8 fragment bases

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    frag_t areg[2][NKS], breg[2][NKS];

    // prologue: tile 0 complete; A0, B0 of tile 1 in flight (its B1, A1 are issued in MFMA-X(0))
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    if (nk > 1) {
        stage(0, 1, 1); stage(2, 1, 1);
        if constexpr (LY) {     // LY: B1 of tile 1 too (only A1 is issued in MFMA-X(0))
            stage(3, 1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FO1_P8_BARRIER();
    if (late) FO1_P8_BARRIER();
    FO1_GEMM_STAMP(1);
    if constexpr ((ABL & 2) != 0 && !FP8) {     // ablation: the fragments are read once, from tile 0
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int fm = 0; fm < 2; ++fm) areg[fm][ks] = *reinterpret_cast<const bf16x8*>(smem + a_off[fm] + (((ks * 2 + hi) ^ a_sw[fm]) << 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) breg[h][ks] = *reinterpret_cast<const bf16x8*>(smem + (2 + h) * GROUP + b_off + (((ks * 2 + hi) ^ b_sw) << 4));
        }
    }

    // STEADY: both later tiles exist — no branch around any DMA piece, wait or barrier (the K loop proper); the last two tiles run the general form
    auto tile = [&](auto BUFC, auto STEADYC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr bool STEADY = decltype(STEADYC)::value;
        const char* base = smem + BUF * BUFSZ;
        const bool more1 = STEADY || t + 1 < nk, more2 = STEADY || t + 2 < nk;
        // FP8: every fragment address derives from ONE register per operand (row * 128 + ((hi * 2) ^ swizzle) * 16): the k-step flips
        // bit 6, the second 16-B slot bit 4, groups / buffers / the second 32-row block are constants.  The empty asm makes the
        // base opaque per use — otherwise hipcc keeps all ~40 derived addresses live in registers and spills (a reload waits
        // vmcnt(0), which also drains the LDS-DMA prefetch: measured 2x slower).
        auto frag8 = [&](int vbase, int imm, int ks) __attribute__((always_inline)) -> v8i32 {
            int t = vbase;
            asm volatile("" : "+v"(t));
            const int a0 = (t ^ (ks << 6)) + imm;
            const uint4 lo = *reinterpret_cast<const uint4*>(smem + a0);
            const uint4 up = *reinterpret_cast<const uint4*>(smem + (a0 ^ 16));
            return v8i32{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
        };
        auto loadA = [&](int h) {
            if constexpr ((ABL & 2) != 0) return;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm) {
                    if constexpr (FP8) areg[fm][ks] = frag8(a_base8, BUF * BUFSZ + h * GROUP + fm * 4096, ks);
                    else areg[fm][ks] = *reinterpret_cast<const bf16x8*>(base + h * GROUP + a_off[fm] + (((ks * 2 + hi) ^ a_sw[fm]) << 4));
                }
        };
        auto loadB = [&](int h) {
            if constexpr ((ABL & 2) != 0) return;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if constexpr (FP8) breg[h][ks] = frag8(b_base8, BUF * BUFSZ + (2 + h) * GROUP, ks);
                else breg[h][ks] = *reinterpret_cast<const bf16x8*>(base + (2 + h) * GROUP + b_off + (((ks * 2 + hi) ^ b_sw) << 4));
            }
        };
        auto mm = [&](const frag_t& b, const frag_t& a, f32x16& c) __attribute__((always_inline)) {
            if constexpr (FP8) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);
        };
        // 16 MFMAs (4 K steps x 2 x 2 fragments) of A half AH with pinned DMA pieces: group gA after the 2nd MFMA of K steps 0 and 1 (LY: of
        // K steps 0 and 2), group gB (if >= 0) after that of K steps 2 and 3
        auto segment = [&](auto AHC, bool on, int gA, int gB, int kt, int buf) __attribute__((always_inline)) {
            constexpr int AH = decltype(AHC)::value;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mm(breg[i >> 1][ks], areg[i & 1][ks], acc[AH * 2 + (i & 1)][i >> 1]);
                    if (NKS == 2) {          // FP8: 8 MFMAs of twice the length — a piece after every second one, group gA first (the waits
                        if (i == 1) pinned(on, ks == 0 ? gA : gB, 0, kt, buf);       // count on gB's pair being the newest)
                        if (i == 3) pinned(on, ks == 0 ? gA : gB, 1, kt, buf);
                    } else if (i == 1) {
                        if (gB >= 0) pinned(on, ks < 2 ? gA : gB, ks & 1, kt, buf);
                        else if (!(ks & 1)) pinned(on, gA, ks >> 1, kt, buf);
                    }
                }
            }
        };
        auto end_load = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 4)) FO1_P8_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto end_mfma = [&](bool last) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 4) && !(last && late)) FO1_P8_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- phase X ----
        loadB(0);
        loadB(1);
        loadA(0);
        // A1 of this tile (issued in MFMA-X(t-1)) must have landed before load-Y; only MFMA-Y(t-1)'s A0, B0 of tile t+1 are newer
        // (LY: behind A1(t) sit A0, B0 [load-Y(t-1)] and B1 [MFMA-Y(t-1)] of tile t+1)
        if constexpr (!(ABL & 8)) {
            if (more1) { if constexpr (LY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        end_load();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
            segment(std::integral_constant<int, 0>{}, !(ABL & 1) && more1, 3, 1, t + 1, BUF ^ 1);
        } else {
            if constexpr (LY) segment(std::integral_constant<int, 0>{}, !(ABL & 1) && more1, 1, -1, t + 1, BUF ^ 1);     // A1(t+1)
            else segment(std::integral_constant<int, 0>{}, !(ABL & 1) && more1, 3, 1, t + 1, BUF ^ 1);
        }
        __builtin_amdgcn_s_setprio(0);
        end_mfma(false);
        // ---- phase Y ----
        loadA(1);
        if constexpr (LY) {
            // A0, B0 of tile t+2 into THIS buffer: both wave halves read those groups in load-X(t), at least one barrier ago.  A0, B0, B1 of
            // tile t+1 must have landed for load-X(t+1): behind them sit A1(t+1) [MFMA-X(t)] and these four pieces
            if (!(ABL & 1) && more2) { stage(0, t + 2, BUF); stage(2, t + 2, BUF); }
            if constexpr (!(ABL & 8)) {
                if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        } else {
            if (!(ABL & 8) && more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        end_load();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
            segment(std::integral_constant<int, 1>{}, !(ABL & 1) && more2, 0, 2, t + 2, BUF);
        } else {
            if constexpr (LY) segment(std::integral_constant<int, 1>{}, !(ABL & 1) && more2, 3, -1, t + 2, BUF);         // B1(t+2)
            else segment(std::integral_constant<int, 1>{}, !(ABL & 1) && more2, 0, 2, t + 2, BUF);
        }
        __builtin_amdgcn_s_setprio(0);
        end_mfma(!more1);
    };
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        tile(std::integral_constant<int, 0>{}, std::true_type{}, t);
        tile(std::integral_constant<int, 1>{}, std::true_type{}, t + 1);
    }
    for (; t < nk; t += 2) {
        tile(std::integral_constant<int, 0>{}, std::false_type{}, t);
        if (t + 1 < nk) tile(std::integral_constant<int, 1>{}, std::false_type{}, t + 1);
    }
    FO1_GEMM_STAMP(2);
    if constexpr (FP8) {
        // dequantise: acc[mf][nf][r] = C[m][n] with m = m_base + mf*32 + (lane & 31), n = n_base + nf*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
        const int mb = m0 + wm * 128 + (lane & 31), nb = n0 + wn * 64 + hi * 4;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int m = mb + mf * 32;
            const float sm = p.scale_m[m < p.M ? m : p.M - 1];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + nf * 32 + g * 8;
                    const float4 sn = *reinterpret_cast<const float4*>(p.scale_n + (n + 4 <= p.N ? n : p.N - 4));
                    acc[mf][nf][g * 4 + 0] *= sm * sn.x; acc[mf][nf][g * 4 + 1] *= sm * sn.y;
                    acc[mf][nf][g * 4 + 2] *= sm * sn.z; acc[mf][nf][g * 4 + 3] *= sm * sn.w;
                }
        }
    }
    if constexpr (EPI == 6 || EPI == 7) {       // fused q/k/v epilogue (fo1_qkv_proj_rope_bf16): always through the LDS image
        epilogue32_qkv<EPI - 6>(p, acc, smem, wave, m0 + wm * 128, n0 + wn * 64, lane);
        FO1_GEMM_STAMP(3);
        return;
    } else {
        if constexpr (EPI != 4) {
            if (p.coal) {   // every wave's LDS reads of the main loop are retired (the last load segment ended at a barrier this wave has passed)
                epilogue32_coalesced<EPI>(p, acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, lane, bz * p.sC, bz * p.sR);
                FO1_GEMM_STAMP(3);
                return;
            }
        }
        epilogue32<EPI, 4, 2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, bz * p.sC, bz * p.sR, blockIdx.z);
    }
}

#ifdef FO1_ENABLE_AB      // persistent tile loop: measured 2-5 % slower, kept for A/B only
// Coalesced epilogue through a 4 KiB wave-private LDS strip (persistent kernel: the 128 KiB image is busy with the next tile):
// the wave's 128 x 64 block in four passes of 32 rows.  Same arithmetic, rounding points and store shapes as epilogue32_coalesced.
template <int EPI>
__device__ __forceinline__ void epilogue32_strip(const GemmParams& p, f32x16 (&acc)[4][2], char* strip, int m_base, int n_base, int lane,
                                                 long long offC, long long offR) {
    const int mi = lane & 31, hi = lane >> 5, hi4 = hi * 4;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        if constexpr (EPI == ACT_SWIGLU16) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const int nb = n_base + nf * 32;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int f0 = g * 8 + hi4;
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float gt = acc[mf][nf][g * 4 + t], up = acc[mf][nf][8 + g * 4 + t];
                        if (p.bias && nb < p.N) { gt += bf16_to_f32(p.bias[nb + f0 + t]); up += bf16_to_f32(p.bias[nb + 16 + f0 + t]); }
                        gt = round_bf16(gt);
                        up = round_bf16(up);
                        o[t] = round_bf16(fo1_silu(gt)) * up;
                    }
                    uint2 ov;
                    ov.x = pack_bf16x2(o[0], o[1]);
                    ov.y = pack_bf16x2(o[2], o[3]);
                    const int q = nf * 2 + g;
                    *reinterpret_cast<uint2*>(strip + mi * 64 + ((q ^ (mi & 3)) << 4) + hi * 8) = ov;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n_out = (n_base >> 1) + (lane & 3) * 8, No = p.N >> 1;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 16 + (lane >> 2), m = m_base + mf * 32 + r;
                const uint4 v = *reinterpret_cast<const uint4*>(strip + r * 64 + (((lane & 3) ^ (r & 3)) << 4));
                if (m < p.M && n_out < No) *reinterpret_cast<uint4*>(p.C + offC + (long long)m * p.ldc + n_out) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip is rewritten by the next pass
        } else {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = n_base + nf * 32 + g * 8 + hi4;
                    float v[4] = {acc[mf][nf][g * 4 + 0], acc[mf][nf][g * 4 + 1], acc[mf][nf][g * 4 + 2], acc[mf][nf][g * 4 + 3]};
                    if (p.bias && n0 < p.N) {
                        const uint2 bv = *reinterpret_cast<const uint2*>(p.bias + n0);
                        v[0] += bf16_lo(bv.x); v[1] += bf16_hi(bv.x); v[2] += bf16_lo(bv.y); v[3] += bf16_hi(bv.y);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = round_bf16(v[t]);
                    if constexpr (EPI != ACT_NONE) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = round_bf16(act_apply_t<EPI>(v[t]));
                    }
                    uint2 ov;
                    ov.x = pack_bf16x2(v[0], v[1]);
                    ov.y = pack_bf16x2(v[2], v[3]);
                    const int q = nf * 4 + g;
                    *reinterpret_cast<uint2*>(strip + mi * 128 + ((q ^ (mi & 7)) << 4) + hi * 8) = ov;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n = n_base + (lane & 7) * 8;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), m = m_base + mf * 32 + r;
                uint4 v = *reinterpret_cast<const uint4*>(strip + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
                if (m < p.M && n < p.N) {
                    if (p.res) {
                        const uint4 rv = *reinterpret_cast<const uint4*>(p.res + offR + (long long)m * p.ldr + n);
                        v.x = pack_bf16x2(bf16_lo(v.x) + bf16_lo(rv.x), bf16_hi(v.x) + bf16_hi(rv.x));
                        v.y = pack_bf16x2(bf16_lo(v.y) + bf16_lo(rv.y), bf16_hi(v.y) + bf16_hi(rv.y));
                        v.z = pack_bf16x2(bf16_lo(v.z) + bf16_lo(rv.z), bf16_hi(v.z) + bf16_hi(rv.z));
                        v.w = pack_bf16x2(bf16_lo(v.w) + bf16_lo(rv.w), bf16_hi(v.w) + bf16_hi(rv.w));
                    }
                    *reinterpret_cast<uint4*>(p.C + offC + (long long)m * p.ldc + n) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip is rewritten by the next pass
        }
    }
}

// ------------------------------------------------------------------------------------------
// PERSISTENT form of gemm_bt_p4_kernel: one workgroup per CU walks its output tiles (virtual id = blockIdx.x + i * gridDim.x in the
// same grouped XCD order) and the K loop simply continues into the next tile: the last two k-tiles of a tile stage the first
// tile and a half of the next one (the DMA source pointers are switched group by group as each group's last use passes), so
// a tile no longer starts with an exposed DMA round trip and its epilogue (stores, residual loads) overlaps those loads.  The
// epilogue therefore cannot use the LDS image: it goes through eight 4 KiB strips beside it (160 KiB of LDS in all).
// Requirements (dispatcher): K / 64 even, no split-K, no batch stride, coalescable C (as epilogue32_coalesced).
// MEASURED (MI355X, profiles/r02_gemm_persistent_ab.log, interleaved A/B on the packed-pass shapes): bit-identical to the one-tile kernel
// and 2-5 % SLOWER on every shape with more than 256 tiles (ViT qkv 947 vs 993 TFLOP/s, LLM gate/up 1111 vs 1142, 8192^3 1293 vs 1350):
// what the hidden prologue gains is lost again — every counted vmcnt wait after the epilogue also waits for the epilogue's own
// stores (one in-order counter for loads and stores), and the chip is already power-limited in the main loop.  NOT the default;
// kept selectable (fo1_gemm_set_big_schedule bit 2) with its bit-identity test.
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bt_p4p_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int GROUP = 16384, BUFSZ = 4 * GROUP;       // A0 | A1 | B0 | B1
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUFSZ] | 8 epilogue strips of 4 KiB

    const int n_tiles = p.tiles_m * p.tiles_n;
    int vt = blockIdx.x;                 // virtual tile id; grid % 8 == 0 keeps every tile of this workgroup in its XCD's run
    int tm, tn;
    tile_coords_grouped_id(p, vt, tm, tn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wave >= 4;
    const long long bz = 0;              // no batch dimension in the persistent form
    const uint16_t* A = p.A + bz * p.sA;
    const uint16_t* W = p.W + bz * p.sW;

    const uint16_t* src[4][2];
    // DMA source pointers of group g (0, 1 = A halves; 2, 3 = W halves) for output tile (tm_, tn_)
    auto set_src = [&](int g, int tm_, int tn_) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = wave * 16 + i * 8 + (lane >> 3);
            const int cs = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;
            if (g < 2) {
                int gm = tm_ * BM + (lr >> 6) * 128 + g * 64 + (lr & 63);
                gm = gm < p.M ? gm : p.M - 1;
                src[g][i] = A + (long long)gm * p.lda + cs;
            } else {
                int gn = tn_ * BN + (lr >> 5) * 64 + (g - 2) * 32 + (lr & 31);
                gn = gn < p.N ? gn : p.N - 1;
                src[g][i] = W + (long long)gn * p.ldw + cs;
            }
        }
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) set_src(g, tm, tn);
    const int nk = p.K / BK;             // even, >= 2 (host-checked); no split-K in the persistent form
    constexpr int kt0 = 0;
    auto stage = [&](int g, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = smem + buf * BUFSZ + g * GROUP + (wave * 2 + i) * 1024;
            const uint16_t* s = (g == 0 ? src[0][i] : g == 1 ? src[1][i] : g == 2 ? src[2][i] : src[3][i]) + (long long)(kt0 + kt) * BK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    const int hi = lane >> 5;
    int a_off[2], a_sw[2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int lr = wm * 64 + fm * 32 + (lane & 31);
        a_off[fm] = lr * 128;
        a_sw[fm] = (lr >> 1) & 7;
    }
    const int b_lr = wn * 32 + (lane & 31);
    const int b_off = b_lr * 128, b_sw = (b_lr >> 1) & 7;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 areg[2][4], breg[2][4];

    // prologue: tile 0 complete; A0, B0 of tile 1 in flight (its B1, A1 are issued in MFMA-X(0))
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    if (nk > 1) {
        stage(0, 1, 1); stage(2, 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FO1_P8_BARRIER();
    if (late) FO1_P8_BARRIER();

    bool has_next = false;
    int tmn = 0, tnn = 0;
    // The k-loop runs on ACROSS output tiles: k-tile nk of this tile is k-tile 0 of the workgroup's next tile, staged by the same
    // schedule (so the next tile starts with its first tile and a half already in LDS, and the hazards are the in-loop ones).
    auto tile = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        const char* base = smem + BUF * BUFSZ;
        const bool more1 = t + 1 < nk || has_next, more2 = t + 2 < nk || has_next;
        const int k1 = t + 1 < nk ? t + 1 : t + 1 - nk, k2 = t + 2 < nk ? t + 2 : t + 2 - nk;
        if (has_next && t == nk - 1) { set_src(1, tmn, tnn); set_src(3, tmn, tnn); }   // MFMA-X stages A1, B1 of the next tile's k-tile 0
        auto loadA = [&](int h) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
                    areg[fm][ks] = *reinterpret_cast<const bf16x8*>(base + h * GROUP + a_off[fm] + (((ks * 2 + hi) ^ a_sw[fm]) << 4));
        };
        auto loadB = [&](int h) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                breg[h][ks] = *reinterpret_cast<const bf16x8*>(base + (2 + h) * GROUP + b_off + (((ks * 2 + hi) ^ b_sw) << 4));
        };
        auto end_load = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            FO1_P8_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto end_mfma = [&](bool last) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(last && late)) FO1_P8_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        };
#define FO1_P4_MFMA4(AH, KS)                                                                                                       \
    acc[(AH) * 2 + 0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[0][KS], areg[0][KS], acc[(AH) * 2 + 0][0], 0, 0, 0);       \
    acc[(AH) * 2 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[0][KS], areg[1][KS], acc[(AH) * 2 + 1][0], 0, 0, 0);       \
    acc[(AH) * 2 + 0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[1][KS], areg[0][KS], acc[(AH) * 2 + 0][1], 0, 0, 0);       \
    acc[(AH) * 2 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[1][KS], areg[1][KS], acc[(AH) * 2 + 1][1], 0, 0, 0);
        // ---- phase X ----
        loadB(0);
        loadB(1);
        loadA(0);
        // A1 of this tile (issued in MFMA-X(t-1)) must have landed before load-Y; only MFMA-Y(t-1)'s A0, B0 of tile t+1 are newer
        if (more1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        end_load();
        __builtin_amdgcn_s_setprio(1);
        FO1_P4_MFMA4(0, 0)
        if (more1) stage(3, k1, BUF ^ 1);
        FO1_P4_MFMA4(0, 1)
        FO1_P4_MFMA4(0, 2)
        if (more1) stage(1, k1, BUF ^ 1);
        FO1_P4_MFMA4(0, 3)
        __builtin_amdgcn_s_setprio(0);
        end_mfma(false);
        // ---- phase Y ----
        if (has_next && t == nk - 2) { set_src(0, tmn, tnn); set_src(2, tmn, tnn); }   // MFMA-Y stages A0, B0 of the next tile from here on
        loadA(1);
        if (more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        end_load();
        __builtin_amdgcn_s_setprio(1);
        FO1_P4_MFMA4(1, 0)
        if (more2) stage(0, k2, BUF);
        FO1_P4_MFMA4(1, 1)
        FO1_P4_MFMA4(1, 2)
        if (more2) stage(2, k2, BUF);
        FO1_P4_MFMA4(1, 3)
        __builtin_amdgcn_s_setprio(0);
        // the lagging half skips the last barrier of EVERY output tile (both halves then run their epilogues side by side, as in
        // the one-tile kernel) and re-enters the stagger with one extra barrier after its epilogue
        end_mfma(t == nk - 1);
#undef FO1_P4_MFMA4
    };
    for (;;) {
        const int vnext = vt + gridDim.x;
        has_next = vnext < n_tiles;
        if (has_next) tile_coords_grouped_id(p, vnext, tmn, tnn);
        for (int t = 0; t < nk; t += 2) {
            tile(std::integral_constant<int, 0>{}, t);
            tile(std::integral_constant<int, 1>{}, t + 1);
        }
        // epilogue through this wave's own 4 KiB strip BESIDE the LDS image (which already holds the next tile's first k-tiles)
        epilogue32_strip<EPI>(p, acc, smem + 2 * BUFSZ + wave * 4096, tm * BM + wm * 128, tn * BN + wn * 64, lane, bz * p.sC, bz * p.sR);
        if (!has_next) break;
        if (late) FO1_P8_BARRIER();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        vt = vnext;
        tm = tmn;
        tn = tnn;
    }
}

#endif   // FO1_ENABLE_AB (gemm_bt_p4p_kernel)

FO1_AB_VAR g_gemm_persist = 0;     // 256x256 kernels: 1 = persistent tile loop with the next tile's first DMA under the epilogue (measured 2-5 % SLOWER than one tile per workgroup, see gemm_bt_p4p_kernel)
FO1_AB_VAR g_gemm_coal = 1;        // 256x256 kernels: LDS-staged coalesced epilogue (0 = fragment-shaped stores, for A/B)
FO1_AB_VAR g_gemm_nt_store = 0;    // 256x256 kernels: non-temporal stores in the coalesced epilogue (A/B)
FO1_AB_VAR g_gemm_big_sched = 1;   // 256x256 kernel schedule: 0 = four phases per K tile (p8), 1 = two fat phases with DMA issued between MFMAs (p4, default: +3..10 % measured, profiles/r02_gemm_bench_p8_v2.log)

// 256 x 256 ping-pong kernel (gemm_bt_p8_kernel)
static int launch_gemm_p8(GemmParams& p, int batch, hipStream_t st) {
    p.tiles_m = cdiv(p.M, 256);
    p.tiles_n = cdiv(p.N, 256);
    if (g_gemm_group_m > 0) p.gm = g_gemm_group_m;
    const dim3 grid(p.tiles_m * p.tiles_n, batch, p.splits);
    const double flops = 2.0 * p.M * (double)p.N * p.K * batch;
    char pname[48];
    const char* name = "gemm_bt_p8<256,256>";
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm %dx%dx%d t256x256 s%d", p.M, p.N, p.K, p.splits);
        name = pname;
    }
    constexpr int smem = 2 * 4 * 16384;
    {
        const int nc = p.act == ACT_SWIGLU16 ? p.N / 2 : p.N;
        p.coal = g_gemm_coal && nc % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && p.sC % 8 == 0 &&
                 (p.res == nullptr || (p.ldr % 8 == 0 && ((uintptr_t)p.res & 15) == 0 && p.sR % 8 == 0));
        if (g_gemm_nt_store != 0 && p.coal) p.coal = 2;
    }
    const int epi = p.splits > 1 ? 4 : p.act;
#ifdef FO1_ENABLE_AB
    if (p.debug & 32) {       // workgroup timeline: the stamps go to the (otherwise unused) split-K partial pointer
        if (p.splits == 1 && g_gemm_stamp_buf && p.coal) p.part = (float*)g_gemm_stamp_buf;
        else p.debug &= ~32;
    }
    static bool attr_done = false;
    if (!attr_done) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int n_tiles = p.tiles_m * p.tiles_n, nk64 = p.K / 64;
    if (g_gemm_big_sched == 1 && g_gemm_persist && p.splits == 1 && batch == 1 && p.coal && n_tiles > 256 && nk64 >= 2 && nk64 % 2 == 0) {
        // persistent form: 256 workgroups (one per CU, a multiple of 8: every workgroup stays in its XCD's run of tiles)
        constexpr int smem_p = 2 * 4 * 16384 + 8 * 4096;
        static bool attrp = false;
        if (!attrp) {
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4p_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4p_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p));
            attrp = true;
        }
        const char* np = (profile_enabled() && g_gemm_profile_shapes) ? name : "gemm_bt_p4p<256,256>";
        const dim3 gp(256, 1, 1);
        if (epi == 0) FO1_LAUNCH(np, flops, gemm_bt_p4p_kernel<0>, gp, dim3(512), smem_p, st, p);
        else if (epi == 1) FO1_LAUNCH(np, flops, gemm_bt_p4p_kernel<1>, gp, dim3(512), smem_p, st, p);
        else if (epi == 2) FO1_LAUNCH(np, flops, gemm_bt_p4p_kernel<2>, gp, dim3(512), smem_p, st, p);
        else FO1_LAUNCH(np, flops, gemm_bt_p4p_kernel<3>, gp, dim3(512), smem_p, st, p);
        return FO1_OK;
    }
#endif
#ifdef FO1_ENABLE_AB
    if (g_gemm_big_sched == 1 && epi == 0 && (p.debug >> 6) & 127) {      // main-loop ablations (debug bits 6-8 = ABL), timing only
        const int abl = (p.debug >> 6) & 127;
#define FO1_ABL_CASE(V)                                                                                                                   \
    case V: {                                                                                                                             \
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<0, false, V>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); \
        FO1_LAUNCH("gemm_bt_p4_ablated", flops, (gemm_bt_p4_kernel<0, false, V>), grid, dim3(512), smem, st, p);                          \
    } break;
        switch (abl) {
            FO1_ABL_CASE(1) FO1_ABL_CASE(2) FO1_ABL_CASE(4) FO1_ABL_CASE(8) FO1_ABL_CASE(16) FO1_ABL_CASE(24) FO1_ABL_CASE(32) FO1_ABL_CASE(64)
            default: return set_err(FO1_ERR_ARG, "gemm: no such ablation %d", abl);
        }
#undef FO1_ABL_CASE
        return FO1_OK;
    }
#endif
    if (g_gemm_big_sched == 1) {
        static bool attr4 = false;
        if (!attr4) {
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr4 = true;
        }
        const char* n4 = (profile_enabled() && g_gemm_profile_shapes) ? name : "gemm_bt_p4<256,256>";
        if (epi == 0) FO1_LAUNCH(n4, flops, gemm_bt_p4_kernel<0>, grid, dim3(512), smem, st, p);
        else if (epi == 1) FO1_LAUNCH(n4, flops, gemm_bt_p4_kernel<1>, grid, dim3(512), smem, st, p);
        else if (epi == 2) FO1_LAUNCH(n4, flops, gemm_bt_p4_kernel<2>, grid, dim3(512), smem, st, p);
        else if (epi == 3) FO1_LAUNCH(n4, flops, gemm_bt_p4_kernel<3>, grid, dim3(512), smem, st, p);
        else FO1_LAUNCH(n4, flops, gemm_bt_p4_kernel<4>, grid, dim3(512), smem, st, p);
    }
#ifdef FO1_ENABLE_AB
    else if (epi == 0) FO1_LAUNCH(name, flops, gemm_bt_p8_kernel<0>, grid, dim3(512), smem, st, p);
    else if (epi == 1) FO1_LAUNCH(name, flops, gemm_bt_p8_kernel<1>, grid, dim3(512), smem, st, p);
    else if (epi == 2) FO1_LAUNCH(name, flops, gemm_bt_p8_kernel<2>, grid, dim3(512), smem, st, p);
    else if (epi == 3) FO1_LAUNCH(name, flops, gemm_bt_p8_kernel<3>, grid, dim3(512), smem, st, p);
    else FO1_LAUNCH(name, flops, gemm_bt_p8_kernel<4>, grid, dim3(512), smem, st, p);
#endif
    if (p.splits > 1) {
        const long long total = (long long)p.M * (p.N / 4);
        const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        FO1_LAUNCH("gemm_splitk_reduce", (double)p.M * p.N * 4.0 * p.splits, gemm_splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
    }
    return FO1_OK;
}

// fp8 form of the 256 x 256 two-phase kernel (fo1_gemm_fp8)
static int launch_gemm_p4_fp8(GemmParams& p, hipStream_t st) {
    p.tiles_m = cdiv(p.M, 256);
    p.tiles_n = cdiv(p.N, 256);
    const dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
    const double flops = 2.0 * p.M * (double)p.N * p.K;
    char pname[56];
    const char* name = "gemm_fp8_p4<256,256>";
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm_fp8 %dx%dx%d t256x256", p.M, p.N, p.K);
        name = pname;
    }
    constexpr int smem = 2 * 4 * 16384;
    {
        const int nc = p.act == ACT_SWIGLU16 ? p.N / 2 : p.N;
        p.coal = g_gemm_coal && nc % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 &&
                 (p.res == nullptr || (p.ldr % 8 == 0 && ((uintptr_t)p.res & 15) == 0));
    }
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    if (p.act == 0) FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<0, true>), grid, dim3(512), smem, st, p);
    else if (p.act == 1) FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<1, true>), grid, dim3(512), smem, st, p);
    else if (p.act == 2) FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<2, true>), grid, dim3(512), smem, st, p);
    else FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<3, true>), grid, dim3(512), smem, st, p);
    return FO1_OK;
}

// One workgroup per row: absmax -> scale = absmax / 448 (1 for an all-zero row) -> q = e4m3(clamp(x / scale, +-448)), RNE (v_cvt_pk_fp8_f32:
// OCP e4m3fn on gfx950).  The same routine quantises activations per token and weights per output channel.
__global__ __launch_bounds__(256) void quantize_rows_e4m3_kernel(const uint16_t* __restrict__ x, long long ldx, int K, uint8_t* __restrict__ q,
                                                                  long long ldq, float* __restrict__ scales) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const uint16_t* xr = x + (long long)row * ldx;
    const int nch = K >> 3;
    constexpr int RC = 6;                       // 16-byte chunks a thread keeps in registers (K <= 12288: the row is read once)
    uint4 keep[RC];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int c = tid + i * 256;
        keep[i] = c < nch ? *reinterpret_cast<const uint4*>(xr + c * 8) : uint4{0, 0, 0, 0};
    }
    auto amax8 = [](const uint4& v, float a) __attribute__((always_inline)) {
        a = fmaxf(a, fmaxf(fmaxf(fabsf(bf16_lo(v.x)), fabsf(bf16_hi(v.x))), fmaxf(fabsf(bf16_lo(v.y)), fabsf(bf16_hi(v.y)))));
        return fmaxf(a, fmaxf(fmaxf(fabsf(bf16_lo(v.z)), fabsf(bf16_hi(v.z))), fmaxf(fabsf(bf16_lo(v.w)), fabsf(bf16_hi(v.w)))));
    };
#pragma unroll
    for (int i = 0; i < RC; ++i) amax = amax8(keep[i], amax);
    for (int c = tid + RC * 256; c < nch; c += 256) amax = amax8(*reinterpret_cast<const uint4*>(xr + c * 8), amax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) s_red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scales[row] = scale;
    uint8_t* qr = q + (long long)row * ldq;
    auto quant8 = [&](const uint4& v, int c) __attribute__((always_inline)) {
        float f[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y), bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i] / scale, -448.0f), 448.0f);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        *reinterpret_cast<uint2*>(qr + c * 8) = uint2{(uint32_t)w0, (uint32_t)w1};
    };
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int c = tid + i * 256;
        if (c < nch) quant8(keep[i], c);
    }
    for (int c = tid + RC * 256; c < nch; c += 256) quant8(*reinterpret_cast<const uint4*>(xr + c * 8), c);   // (L2 hit: just read)
}

template <int BM, int BN, int NS, int WGM = 2, int WGN = 2>
static int launch_ring(GemmParams& p, const char* name, double flops, dim3 grid, hipStream_t st) {
    static_assert((NS - 2) * ((BM + BN) / 32) <= 63, "vmcnt is a 6-bit counter");
    static_assert(NS - 2 <= 4, "the K loop's counted waits cover up to 4 tiles in flight behind the current one");
    constexpr int smem = NS * (BM + BN) * 128;
    static_assert(smem <= 160 * 1024, "LDS per workgroup");
    static bool attr_done = false;
    if (!attr_done) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_ring_kernel<BM, BN, NS, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    FO1_LAUNCH(name, flops, (gemm_bt_ring_kernel<BM, BN, NS, WGM, WGN>), grid, dim3(256), smem, st, p);
    return FO1_OK;
}

// Skinny-M weight streams of the decode pool (65..128 rows; round 6): tiles whose ring is DEEP — what a weight stream needs is bytes in flight
// per CU (HBM latency x 6 TB/s = ~48 KB per CU chip-wide), and a 128 x 128 tile with 3 stages keeps 32 KB of W in flight on only 172 CUs.
//   <128, 96> waves 4 x 1, 5 stages (140 KB): 48 KB of W in flight per workgroup, 230 workgroups for N = 22016 (SwiGLU pairs stay in a wave);
//   <128, 64> waves 2 x 2, 6 stages (144 KB): 40 KB of W in flight, N / 64 column tiles x splits workgroups (down as split-K planes);
//   <128, 64> 3 stages (72 KB): two workgroups per CU.
template <int BM, int BN, int WGM, int WGN>
static int launch_ring_deep(GemmParams& p, int ns, hipStream_t st) {
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(p.N, BN);
    const dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
    const double flops = 2.0 * p.M * (double)p.N * p.K;
    char pname[56];
    snprintf(pname, sizeof pname, "gemm_bt_ring<%d,%d,%d>", BM, BN, ns);
    if (profile_enabled() && g_gemm_profile_shapes) snprintf(pname, sizeof pname, "gemm %dx%dx%d t%dx%d s%d r%d", p.M, p.N, p.K, BM, BN, p.splits, ns);
    p.stages = ns;
    if constexpr (BN == 96) {
        if (ns == 3) return launch_ring<BM, BN, 3, WGM, WGN>(p, pname, flops, grid, st);
        if (ns == 4) return launch_ring<BM, BN, 4, WGM, WGN>(p, pname, flops, grid, st);
        return launch_ring<BM, BN, 5, WGM, WGN>(p, pname, flops, grid, st);
    } else {
        if (ns == 3) return launch_ring<BM, BN, 3, WGM, WGN>(p, pname, flops, grid, st);
        if (ns == 4) return launch_ring<BM, BN, 4, WGM, WGN>(p, pname, flops, grid, st);
        return launch_ring<BM, BN, 6, WGM, WGN>(p, pname, flops, grid, st);
    }
}

template <int BM, int BN>
static int launch_gemm(GemmParams& p, int batch, bool glds, hipStream_t st, bool reduce = true) {
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(p.N, BN);
    const dim3 grid(p.tiles_m * p.tiles_n, batch, glds ? p.splits : 1);
    const double flops = 2.0 * p.M * (double)p.N * p.K * batch;
    char pname[48];
    // one profile row per kernel template, so the rows map 1:1 onto rocprofv3's kernel names
    const char* name = !glds ? "gemm_bf16_reg"
                     : (BM == 128 ? "gemm_bt_glds<128,128>" : (BN == 128 ? "gemm_bt_glds<64,128>" : "gemm_bt_glds<64,64>"));
    if (glds && p.stages >= 3) {
        const int ns = p.stages == 3 ? 3 : ((p.stages == 4 || BM + BN > 128) ? 4 : 6);
        snprintf(pname, sizeof pname, "gemm_bt_ring<%d,%d,%d>", BM, BN, ns);
        name = pname;
    }
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm %dx%dx%d t%dx%d s%d r%d", p.M, p.N, p.K, BM, BN, glds ? p.splits : 1, glds ? p.stages : 0);
        name = pname;
    }
    if (glds && p.stages >= 3) {
        int rc = FO1_OK;
        if (p.stages == 3) rc = launch_ring<BM, BN, 3>(p, name, flops, grid, st);
        else if (p.stages == 4) rc = launch_ring<BM, BN, 4>(p, name, flops, grid, st);
        else if constexpr (BM + BN <= 128) rc = launch_ring<BM, BN, 6>(p, name, flops, grid, st);
        else rc = launch_ring<BM, BN, 4>(p, name, flops, grid, st);
        if (rc != FO1_OK) return rc;
        if (p.splits > 1 && reduce) {
            const long long total = (long long)p.M * (p.N / 4);
            const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            FO1_LAUNCH("gemm_splitk_reduce", (double)p.M * p.N * 4.0 * p.splits, gemm_splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
        }
        return FO1_OK;
    }
    if (glds) {
        constexpr int smem = 2 * (BM + BN) * 128;
        static bool attr_done = false;
        if (!attr_done) {
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_glds_kernel<BM, BN>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_done = true;
        }
        FO1_LAUNCH(name, flops, (gemm_bt_glds_kernel<BM, BN>), grid, dim3(256), smem, st, p);
        if (p.splits > 1 && reduce) {
            const long long total = (long long)p.M * (p.N / 4);
            const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            FO1_LAUNCH("gemm_splitk_reduce", (double)p.M * p.N * 4.0 * p.splits, gemm_splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
        }
    } else {
        FO1_LAUNCH(name, flops, (gemm_bt_reg_kernel<BM, BN>), grid, dim3(256), 0, st, p);
    }
    return FO1_OK;
}

// q/k/v projection with the fused epilogue (EPI 6 LLM / 7 ViT): always the 256 x 256 two-phase kernel (the epilogue pairs waves of one tile row)
static int launch_qkv_p4(GemmParams& p, int mode, hipStream_t st) {
    p.tiles_m = cdiv(p.M, 256);
    p.tiles_n = cdiv(p.N, 256);
    if (g_gemm_group_m > 0) p.gm = g_gemm_group_m;
    p.splits = 1;
    p.kper = p.K / 64 + 1;
    p.part = nullptr;
    p.debug = 0;
#ifdef FO1_ENABLE_AB
    if ((g_gemm_debug & 32) && g_gemm_stamp_buf) { p.debug = 32; p.part = (float*)g_gemm_stamp_buf; }      // workgroup timeline (scripts/r06_qkv_timeline.py)
#endif
    p.coal = 1;
    p.stages = 2;
    const dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
    const double flops = 2.0 * p.M * (double)p.N * p.K;
    constexpr int smem = 2 * 4 * 16384;
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    char pname[56];
    const char* name = "gemm_bt_p4<256,256>";
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm %dx%dx%d t256x256 qkv%d", p.M, p.N, p.K, mode);
        name = pname;
    }
    if (mode == 0) FO1_LAUNCH(name, flops, gemm_bt_p4_kernel<6>, grid, dim3(512), smem, st, p);
    else FO1_LAUNCH(name, flops, gemm_bt_p4_kernel<7>, grid, dim3(512), smem, st, p);
    return FO1_OK;
}

// THE statement of "this [M, K] x [N, K]^T product runs on the 256 x 256 two-phase kernel": gemm_dispatch picks its tile with it and
// fo1_gemm_takes_big_tile answers callers with it (one predicate, ADVICE r5: the fused q/k/v and implicit-convolution forms are bit-identical to
// the two-launch forms only while both sides agree).  Inputs beyond the shape: the staging pin (variant 1 = register staging has no 256 x 256
// form), the tile pin, a forced split-K (the fused epilogues have no split-K form; test / bench build only — constants in the product).
// large M (batched prefill): the kernel is taken once its tiles fill >= 60 % of the rounds they occupy and at least half the CUs (measured,
// profiles/r02_gemm_bench_p8_v1.log: LLM o/down at 168 tiles 760 / 1007 TF vs 667 / 684 for the best small tile; merger 260 tiles = 51 % of two
// rounds loses, 659 vs 894); K >= 256 (nk >= 4): with the two-phase schedule and the coalesced epilogue even four K tiles per output tile beat
// the 64 x 128 tile (DaViT stage 0 at 25 images: ~400 vs 211 TFLOP/s).
static inline bool big_tile_rule(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 != 0 || g_gemm_variant == 1) return false;
    if (g_gemm_tile != 0) return g_gemm_tile == 5;
    const int nk = K / 64;
    const long long t256 = (long long)cdiv(M, 256) * cdiv(N, 256) * batch;
    return nk >= 4 && M >= 1024 && t256 >= 128 && (double)t256 / (double)(cdiv((int)t256, 256) * 256) >= 0.6;
}

int gemm_dispatch(GemmParams& p, int batch, hipStream_t st, float* ws, size_t ws_bytes) {
    bool glds = (p.K % 64 == 0);
    if (g_gemm_variant == 1) glds = false;
    if (g_gemm_variant >= 2 && p.K % 64 != 0) return set_err(FO1_ERR_ARG, "gemm: glds variant needs K %% 64 == 0 (K=%d)", p.K);
    // Dispatch heuristics measured on MI355X with COLD weights (scripts/gemm_bench.py cold,
    // profiles/r01_gemm_bench_cold.log): in the pipeline every GEMM streams its weights from HBM (8 GB of weights
    // per step against a 256 MB Infinity Cache), which a warm micro-benchmark hides.
    //  * 128x128 tiles once they alone give >= 3 workgroups per CU, else 64x128 if that gives >= 2 per CU,
    //    else 64x64; 128x256 (8 waves) for >= 4 full rounds of such tiles;
    //  * few-tile launches (<= 3 workgroups per CU) are latency-bound on HBM misses: they take the 3-deep
    //    LDS ring (two k-tiles in flight per workgroup): LLM qkv 25.9 -> 18.2 us, ViT down 39.3 -> 29.2 us;
    //    many-tile launches keep the two-stage kernel, whose smaller LDS footprint holds more workgroups per CU;
    //  * split-K only for deep-K skinny outputs (K >= 4096), to about one workgroup per CU:
    //    LLM down projection 515x2048x11008: 131 us unsplit -> 52 us with 64x128 tiles x 2 splits on the ring.
    const long long t128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128) * batch;
    const long long t64x128 = (long long)cdiv(p.M, 64) * cdiv(p.N, 128) * batch;
    const long long t64 = (long long)cdiv(p.M, 64) * cdiv(p.N, 64) * batch;
    const int nk = p.K / 64;
    int tile = g_gemm_tile;
    int splits = g_gemm_splitk;
    const bool can_split = glds && batch == 1 && ws != nullptr && p.N % 4 == 0 && p.act != ACT_SWIGLU16;
    const bool auto_tile = tile == 0;
    if (auto_tile) {
        tile = (t128 >= 768 && nk >= 16) ? 1 : (t64x128 >= 512 ? 2 : 3);   // shallow K (DaViT stage 0, K=256): 64x128 wins
        if (glds && nk >= 16 && (long long)cdiv(p.M, 128) * cdiv(p.N, 256) * batch >= 1024) tile = 4;
        if (splits == 0 && can_split && nk >= 64 && t64x128 < 512) tile = 2;
        // 64 < M <= 128 with >= 128 row tiles (the decode pool's gate/up and lm_head products, llm.DecodePool): a weight stream whose
        // activations come back from L2 once per tile column — 128 x 128 tiles halve that re-read against 64 x 64 (cold weights,
        // profiles/r04_pool_gemm_stream_kernel_vs_tile_kernels.json: gate/up 34.4 -> 30.5 us, lm_head 163 -> 152 us at M = 128)
        if (glds && p.M > 64 && p.M <= 128 && t128 >= 128 && nk >= 16) tile = 1;
        // ... and the interleaved gate/up product (SwiGLU epilogue, N = 22016 -> 230 tiles of 96 columns = one per CU where 172 tiles of 128 leave a
        // third of the chip idle): <128, 96>, waves 4 x 1 so that the [gate 16 | up 16] pairs stay inside a wave, 3 ring stages — deeper rings
        // measured no faster (a CU's fetch rate is capped by its outstanding requests, not by the bytes a ring keeps in flight).  28.4 against
        // 32.5-33.0 us at M = 128 with cold weights (profiles/r06_pool_gemm_deep_ring_tiles.json); same K order per element: bit-identical output
        if (glds && p.M > 64 && p.M <= 128 && nk >= 16 && p.act == ACT_SWIGLU16 && p.N >= 8192 && batch == 1 && p.C32 == nullptr) tile = 6;
        if (big_tile_rule(p.M, p.N, p.K, batch)) tile = 5;      // large M (batched prefill): the 256 x 256 two-phase kernel
    }
    p.splits = 1;
    p.kper = nk + 1;
    p.part = nullptr;
    p.debug = g_gemm_debug;
    long long tiles = tile == 3 ? t64 : (tile == 2 ? t64x128 : (tile == 7 ? (long long)cdiv(p.M, 128) * cdiv(p.N, 64) : t128));
    if (can_split && tile != 6) {
        if (splits == 0) {
            splits = 1;
            if (nk >= 64 && tiles < 512 && tile == 2) {
                splits = (int)((256 + tiles - 1) / tiles);
                if (splits > 8) splits = 8;
            }
        }
        if (splits > nk) splits = nk;
        while (splits > 1 && (size_t)splits * p.M * p.N * sizeof(float) > ws_bytes) --splits;
        if (splits > 1) {
            p.kper = cdiv(nk, splits);
            p.splits = cdiv(nk, p.kper);
            p.part = ws;
        }
    }
    const bool p8_ok = glds && p.C32 == nullptr && p.N % 4 == 0 && p.ldc % 4 == 0 && ((uintptr_t)p.C & 7) == 0 && p.sC % 4 == 0 &&
                       (p.bias == nullptr || ((uintptr_t)p.bias & 7) == 0) &&
                       (p.res == nullptr || (p.ldr % 4 == 0 && ((uintptr_t)p.res & 7) == 0 && p.sR % 4 == 0)) && (p.act != ACT_SWIGLU16 || p.N % 32 == 0) && p.act <= ACT_SWIGLU16;   // (ReLU: the smaller tiles' run-time activation)
    if (tile == 5 && p8_ok) {
        p.stages = 2;
        return launch_gemm_p8(p, batch, st);
    }
    if (tile == 5) tile = 1;
    if ((tile == 6 || tile == 7) && !(glds && batch == 1 && p.C32 == nullptr)) tile = 1;
    if (tile == 6) return launch_ring_deep<128, 96, 4, 1>(p, g_gemm_variant >= 3 ? g_gemm_variant : 3, st);
    if (tile == 7) {
        const int rc = launch_ring_deep<128, 64, 2, 2>(p, g_gemm_variant >= 3 ? g_gemm_variant : (p.splits > 1 ? 6 : 3), st);
        if (rc != FO1_OK) return rc;
        if (p.splits > 1) {
            const long long total = (long long)p.M * (p.N / 4);
            const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            FO1_LAUNCH("gemm_splitk_reduce", (double)p.M * p.N * 4.0 * p.splits, gemm_splitk_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
        }
        return FO1_OK;
    }
    if (g_gemm_variant >= 3) p.stages = g_gemm_variant;
    else if (g_gemm_variant == 0 && glds && ((tile == 3 && t64 <= 768) || (tile == 2 && tiles * p.splits < 512) || (tile == 1 && auto_tile && p.M <= 128))) p.stages = 3;
    else p.stages = 2;
    if (tile == 4 && glds && p.stages == 2) return launch_gemm_wide<128, 256>(p, batch, st);
    if (tile == 4) tile = 1;
    if (tile == 1) return launch_gemm<128, 128>(p, batch, glds, st);
    if (tile == 2) return launch_gemm<64, 128>(p, batch, glds, st);
    return launch_gemm<64, 64>(p, batch, glds, st);
}

}  // namespace fo1

extern "C" {

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: test / bench build only
int fo1_gemm_set_variant(int staging, int tile) {
    if (staging < 0 || staging > 6 || tile < 0 || tile > 7 || (staging == 5 && tile != 6)) return fo1::set_err(FO1_ERR_ARG, "gemm: bad variant %d/%d", staging, tile);
    fo1::g_gemm_variant = staging;
    fo1::g_gemm_tile = tile;
    return FO1_OK;
}

int fo1_gemm_set_big_schedule(int sched) {
    // bit 0: 0 = four phases per K tile, 1 = two fat phases;  bit 1 set = fragment-shaped (un-coalesced) epilogue stores;
    // bit 2 set = persistent tile loop (gemm_bt_p4p_kernel) instead of one output tile per workgroup
    if (sched < 0 || sched > 15) return fo1::set_err(FO1_ERR_ARG, "gemm: bad 256x256 schedule %d", sched);
    fo1::g_gemm_nt_store = (sched & 8) ? 1 : 0;      // bit 3: non-temporal epilogue stores (A/B)
    fo1::g_gemm_big_sched = sched & 1;
    fo1::g_gemm_coal = (sched & 2) ? 0 : 1;
    fo1::g_gemm_persist = (sched & 4) ? 1 : 0;
    return FO1_OK;
}

int fo1_gemm_set_splitk(int splits) {
    if (splits < 0 || splits > 64) return fo1::set_err(FO1_ERR_ARG, "gemm: bad split-K %d", splits);
    fo1::g_gemm_splitk = splits;
    return FO1_OK;
}

int fo1_gemm_set_debug(int bits) {
    fo1::g_gemm_debug = bits;
    return FO1_OK;
}

int fo1_gemm_set_stamp_buffer(void* device_buffer) {
    fo1::g_gemm_stamp_buf = device_buffer;
    return FO1_OK;
}

int fo1_gemm_set_group_m(int rows) {
    if (rows < 0 || rows > 1024) return fo1::set_err(FO1_ERR_ARG, "gemm_set_group_m: %d", rows);
    fo1::g_gemm_group_m = rows;
    return FO1_OK;
}

int fo1_gemm_set_gemv(int on) {
    fo1::g_gemm_gemv = on != 0;
    return FO1_OK;
}
#endif   // FO1_ENABLE_AB

#ifdef FO1_ENABLE_AB      // instrument (include/fo1_ab.h)
int fo1_gemm_profile_shapes(int on) {
    fo1::g_gemm_profile_shapes = on != 0;
    fo1::g_gemv_profile_shapes = on != 0;
    return FO1_OK;
}
#endif

int fo1_gemm_bf16_ws(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                     void* C, int ldc, int M, int N, int K, int act, int out_f32, void* workspace, size_t workspace_bytes,
                     void* stream);

int fo1_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                  void* C, int ldc, int M, int N, int K, int act, int out_f32, void* stream) {
    return fo1_gemm_bf16_ws(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, out_f32, nullptr, 0, stream);
}

int fo1_gemm_bf16_ws(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                     void* C, int ldc, int M, int N, int K, int act, int out_f32, void* workspace, size_t workspace_bytes,
                     void* stream) {
    using namespace fo1;
    if (M == 0 || N == 0) return FO1_OK;
    FO1_CHECK_ARG(A && W && C, "gemm: NULL operand");
    FO1_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    FO1_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm: K, lda, ldw must be multiples of 8 (K=%d lda=%d ldw=%d)", K, lda, ldw);
    FO1_CHECK_ARG(lda >= K && ldw >= K, "gemm: leading dimension too small");
    FO1_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: A/W must be 16-byte aligned");
    FO1_CHECK_ARG((act >= 0 && act <= 3) || act == 5, "gemm: act=%d (0 none, 1 GELU, 2 SiLU, 3 interleaved SwiGLU, 5 ReLU)", act);
    if (act == 3) {
        FO1_CHECK_ARG(!out_f32 && residual == nullptr && N % 32 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0,
                      "gemm: swiglu epilogue needs bf16 out, no residual, N %% 32 == 0 (N=%d), ldc %% 4 == 0", N);
        FO1_CHECK_ARG(ldc >= N / 2, "gemm: swiglu output has N/2 columns; ldc too small");
    } else {
        FO1_CHECK_ARG(ldc >= N, "gemm: ldc too small");
    }
    FO1_CHECK_ARG(!out_f32 || residual == nullptr, "gemm: fp32 output does not take a residual");
    FO1_CHECK_ARG(residual == nullptr || ldr >= N, "gemm: ldr too small");
    GemmParams p;
    p.A = (const uint16_t*)A; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual;
    p.C = out_f32 ? nullptr : (uint16_t*)C;
    p.C32 = out_f32 ? (float*)C : nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.coal = 0;
    p.scale_m = p.scale_n = nullptr;
    FO1_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "gemm: workspace must be 16-byte aligned");
    if (g_gemm_gemv && M <= 4 && !out_f32 && (size_t)(M > 2 ? 4 : M) * K * 2 <= 150 * 1024 && (act != 3 || N % 32 == 0))
        return gemv_dispatch(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, (hipStream_t)stream, nullptr, 0.f);
    return gemm_dispatch(p, 1, (hipStream_t)stream, (float*)workspace, workspace_bytes);
}

// 3x3 convolution as an IMPLICIT GEMM on the 256 x 256 kernel (round 5, VERDICT r4 #1b; DaViT ConvEmbed modeling_davit.py:102-148, SimpleFPN
// simple_fpn.py:141-176): C[m, :] = sum over (ky, kx, c) of Xpad[row(m) + (ky, kx)][c] * W[:, ky, kx, c] + bias — what fo1_im2col_bf16 +
// fo1_gemm_bf16 compute (same kernel, same K order: bit-identical) without the [M, 9 Cin] column matrix (2.2 GB per launch at the FPN's
// finest level).  Xpad: token-major map zero-padded by one pixel per side, row pitch Wp pixels (written there by fo1_layernorm_rows_bf16);
// a_rows [M] uint32: byte offset in Xpad of output pixel m's top-left tap (host-built: any stride, any batch of images sharing Wp);
// W [N][3][3][Cin] bf16, Cin = 64 * 2^j.  act 0 / 1 (GELU).
int fo1_conv3x3_gemm_bf16(const void* Xpad, const uint32_t* a_rows, int Wp, int Cin, const void* W, int ldw, const void* bias, void* C, int ldc, int M,
                          int N, int act, void* stream) {
    using namespace fo1;
    if (M == 0) return FO1_OK;
    FO1_CHECK_ARG(Xpad && a_rows && W && C, "conv3x3_gemm: NULL operand");
    FO1_CHECK_ARG(Cin >= 64 && (Cin & (Cin - 1)) == 0 && Wp >= 3, "conv3x3_gemm: Cin=%d must be a power of two >= 64 (Wp=%d)", Cin, Wp);
    FO1_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && ldw % 8 == 0 && ldw >= 9 * Cin && ldc % 8 == 0 && ldc >= N, "conv3x3_gemm: M=%d N=%d ldw=%d ldc=%d", M, N, ldw, ldc);
    FO1_CHECK_ARG(((uintptr_t)Xpad & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 7) == 0),
                  "conv3x3_gemm: operands must be 16-byte aligned");
    FO1_CHECK_ARG(act == 0 || act == 1, "conv3x3_gemm: act=%d (0 none, 1 GELU)", act);
    GemmParams p;
    p.A = (const uint16_t*)Xpad; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = nullptr;
    p.C = (uint16_t*)C; p.C32 = nullptr;
    p.M = M; p.N = N; p.K = 9 * Cin; p.lda = Cin; p.ldw = ldw; p.ldc = ldc; p.ldr = 0; p.act = act;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.scale_m = p.scale_n = nullptr;
    p.a_rows = a_rows; p.conv_row_bytes = (uint32_t)Wp * Cin * 2u; p.conv_cin_bytes = (uint32_t)Cin * 2u;
    int lg = 0;
    while ((64 << lg) < Cin) ++lg;
    p.conv_lgc = lg;
    p.tiles_m = cdiv(M, 256);
    p.tiles_n = cdiv(N, 256);
    if (g_gemm_group_m > 0) p.gm = g_gemm_group_m;
    p.splits = 1; p.kper = p.K / 64 + 1; p.part = nullptr; p.debug = 0; p.coal = 1; p.stages = 2;
    const dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
    constexpr int smem = 2 * 4 * 16384;
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<0, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bt_p4_kernel<1, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const double flops = 2.0 * M * (double)N * p.K;
    char pname[56];
    const char* name = "gemm_bt_p4<256,256>";
    if (profile_enabled() && g_gemm_profile_shapes) {
        snprintf(pname, sizeof pname, "gemm %dx%dx%d t256x256 conv", M, N, p.K);
        name = pname;
    }
    hipStream_t st = (hipStream_t)stream;
    if (act == 0) FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<0, false, 0, true>), grid, dim3(512), smem, st, p);
    else FO1_LAUNCH(name, flops, (gemm_bt_p4_kernel<1, false, 0, true>), grid, dim3(512), smem, st, p);
    return FO1_OK;
}

// 1 when fo1_gemm_bf16 runs an [M, K] x [N, K]^T product (bf16 out, K % 64 == 0, aligned operands) on the 256 x 256 two-phase kernel — the
// rule of gemm_dispatch (and, in the test / bench build, its tile pin).  fo1_qkv_proj_rope_bf16 always runs on that kernel: a caller that
// wants the fused and the two-launch form to agree BIT FOR BIT takes the fused one exactly where this says 1.
int fo1_gemm_takes_big_tile(int M, int N, int K) {
    using namespace fo1;
    return (big_tile_rule(M, N, K, 1) && g_gemm_splitk <= 1) ? 1 : 0;      // (a forced split-K — test / bench build — sends the plain GEMM through fp32 planes)
}

// q/k/v projection + bias + rotary embedding + K-cache append + V^T write in ONE launch (the 256 x 256 kernel with the fused epilogue
// epilogue32_qkv): what fo1_gemm_bf16 followed by fo1_qkv_post_llm_bf16 / fo1_qkv_post_vit_bf16 computes, bit for bit, without the second pass
// over the [M, N] activation.  mode 0 (LLM, modeling_qwen2_5_vl.py:643-685): W rows [q heads | k heads | v heads], head_dim 128; cos / sin bf16
// [M][128]; q (rotated) -> C[:, :n_q * 128]; k (rotated) -> kcache[kv head][pos0 + m][128]; v -> vt[kv head * 128 + d][pos0 + m]; the k / v columns of
// C are NOT written.  mode 1 (ViT, :219-230, :162-169): W rows HEAD-MAJOR, per head [q 80 | k 80 | v 80 | 16 zero rows] (N = 256 * n_q_heads); cos /
// sin fp32 [M][40]; q, k (rotated) -> C in that layout (attention: head stride 256, k at column 80); v -> vt[head * 80 + d][pos0 + m].
// K % 64 == 0, N % 256 == 0, pos0 % 8 == 0, vt_ld % 8 == 0, 16-byte aligned operands.
int fo1_qkv_proj_rope_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K, int mode,
                           int n_q_heads, int n_kv_heads, const void* cos_table, const void* sin_table, void* kcache, long long kcache_head_stride,
                           int pos0, void* vt, long long vt_ld, void* stream) {
    using namespace fo1;
    if (M == 0) return FO1_OK;
    FO1_CHECK_ARG(A && W && C && cos_table && sin_table && vt, "qkv_proj_rope: NULL operand");
    FO1_CHECK_ARG(mode == 0 || mode == 1, "qkv_proj_rope: mode %d (0 LLM, 1 ViT head-major)", mode);
    FO1_CHECK_ARG(M > 0 && K >= 128 && K % 64 == 0 && N % 256 == 0, "qkv_proj_rope: M=%d N=%d (%% 256) K=%d (%% 64, >= 128)", M, N, K);
    FO1_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ldc % 8 == 0 && ldc >= (mode == 0 ? n_q_heads * 128 : N), "qkv_proj_rope: leading dimensions");
    FO1_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)vt & 15) == 0 &&
                  ((uintptr_t)cos_table & 15) == 0 && ((uintptr_t)sin_table & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 7) == 0),
                  "qkv_proj_rope: operands must be 16-byte aligned");
    FO1_CHECK_ARG(pos0 >= 0 && pos0 % 8 == 0 && vt_ld % 8 == 0 && vt_ld >= pos0 + M, "qkv_proj_rope: pos0=%d (%% 8), vt_ld=%lld (%% 8, >= pos0 + M)", pos0, vt_ld);
    if (mode == 0) {
        FO1_CHECK_ARG(n_q_heads > 0 && n_kv_heads > 0 && N == (n_q_heads + 2 * n_kv_heads) * 128, "qkv_proj_rope: N=%d is not (%d + 2 x %d) heads of 128", N, n_q_heads, n_kv_heads);
        FO1_CHECK_ARG(kcache && ((uintptr_t)kcache & 15) == 0 && kcache_head_stride % 8 == 0, "qkv_proj_rope: kcache");
    } else {
        FO1_CHECK_ARG(n_q_heads > 0 && N == n_q_heads * 256, "qkv_proj_rope: N=%d is not %d head tiles of 256", N, n_q_heads);
    }
    GemmParams p;
    p.A = (const uint16_t*)A; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = nullptr;
    p.C = (uint16_t*)C; p.C32 = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = 0; p.act = 0;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.scale_m = p.scale_n = nullptr;
    p.rope_cos = cos_table; p.rope_sin = sin_table;
    p.kcache = (uint16_t*)kcache; p.kc_head_stride = kcache_head_stride;
    p.vt = (uint16_t*)vt; p.vt_ld = vt_ld; p.pos0 = pos0; p.n_q = n_q_heads; p.n_kv = n_kv_heads;
    return launch_qkv_p4(p, mode, (hipStream_t)stream);
}

// Split-K partial sums only (the decode pool's q/k/v, o and down projections, llm.DecodePool): part[z][m][n] (fp32, row stride N) = the
// product over the z-th run of K tiles, z < *splits_out = the effective split count for the request (K tiles of 64 dealt in equal runs,
// the last may be shorter).  No epilogue and no reduce launch: the consumer sums the planes in z order — fo1_splitk_residual_rmsnorm_bf16
// (+ residual, + the next RMSNorm) or fo1_pool_qkv_post_partials_bf16 (+ bias, RoPE, cache append).  LDS-DMA ring kernel, 64 x 64 tiles
// (64 x 128 when those alone give a workgroup per CU).
int fo1_gemm_bf16_partials(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int splits, float* part, int* splits_out,
                           void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(A && W && part && splits_out, "gemm_partials: NULL operand");
    FO1_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0, "gemm_partials: bad shape M=%d N=%d K=%d (K %% 64, N %% 4)", M, N, K);
    FO1_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "gemm_partials: lda / ldw");
    FO1_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)part & 15) == 0, "gemm_partials: operands must be 16-byte aligned");
    const int nk = K / 64;
    FO1_CHECK_ARG(splits >= 2 && splits <= nk && splits <= 64, "gemm_partials: splits=%d (2 .. min(64, K / 64 = %d))", splits, nk);
    GemmParams p;
    p.A = (const uint16_t*)A; p.W = (const uint16_t*)W; p.bias = nullptr; p.res = nullptr; p.C = nullptr; p.C32 = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N; p.ldr = 0; p.act = ACT_NONE;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.coal = 0; p.debug = 0; p.stages = 3;
    p.scale_m = p.scale_n = nullptr;
    p.kper = cdiv(nk, splits);
    p.splits = cdiv(nk, p.kper);
    p.part = part;
    *splits_out = p.splits;
    if (p.splits < 2) return set_err(FO1_ERR_ARG, "gemm_partials: K too shallow for %d splits", splits);
    // wide outputs (gate/up: 22016 columns) at 65..128 rows: 128 x 256 tiles — the activations come back from L2 once per tile COLUMN, and
    // 86 column tiles x 3 planes fill the chip where 172 tiles of 128 x 128 with the SwiGLU epilogue leave a third of it idle
    if (g_gemm_tile == 7 && M > 64 && M <= 128) return launch_ring_deep<128, 64, 2, 2>(p, g_gemm_variant >= 3 ? g_gemm_variant : 6, (hipStream_t)stream);   // (A/B pin)
    if (M > 64 && M <= 128 && N >= 8192 && nk >= 16) return launch_gemm_wide<128, 256>(p, 1, (hipStream_t)stream, false);
    // (128 x 128 tiles for the 65..128-row down projection — the weights fetched once instead of once per 64-row tile — measured no faster:
    // 17.4 vs 16.3 us at 12-16 planes, profiles/r04_pool_step_splitk_sweep.json)
    if ((long long)cdiv(M, 64) * cdiv(N, 128) * p.splits >= 256) return launch_gemm<64, 128>(p, 1, true, (hipStream_t)stream, false);
    return launch_gemm<64, 64>(p, 1, true, (hipStream_t)stream, false);
}

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: a measured no-gain form, test / bench build only
// fo1_gemm_bf16 for a weight streamed ONCE per call by few rows (the decode pool's gate/up and lm_head at 65..128 rows, 33..64 rows on the
// 64 x 128 tile): W_tiled is the copy of W [N, K] laid out [N / 128][K / 64][128][64] (ops.tile_weight), so that every K tile of a column tile is
// one contiguous 16 KB block instead of 128 pieces of 128 B at a 2 K-byte stride.  Same kernel (LDS-DMA ring, BN = 128), same arithmetic and
// epilogues — results bit-identical to fo1_gemm_bf16 on the row-major W when that call takes the same tile.  N % 128 == 0, K % 64 == 0, M <= 128.
int fo1_gemm_bf16_wtiled(const void* A, int lda, const void* W_tiled, const void* bias, const void* residual, int ldr, void* C, int ldc, int M, int N,
                         int K, int act, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(A && W_tiled && C, "gemm_wtiled: NULL operand");
    FO1_CHECK_ARG(M >= 1 && M <= 128 && N >= 128 && N % 128 == 0 && K >= 64 && K % 64 == 0, "gemm_wtiled: M=%d (<= 128) N=%d (%% 128) K=%d (%% 64)", M, N, K);
    FO1_CHECK_ARG(lda % 8 == 0 && lda >= K && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W_tiled & 15) == 0, "gemm_wtiled: A / W alignment");
    FO1_CHECK_ARG((act >= 0 && act <= 3) || act == 5, "gemm_wtiled: act=%d", act);
    if (act == 3) FO1_CHECK_ARG(residual == nullptr && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0 && ldc >= N / 2, "gemm_wtiled: swiglu epilogue layout");
    else FO1_CHECK_ARG(ldc >= N, "gemm_wtiled: ldc too small");
    FO1_CHECK_ARG(residual == nullptr || ldr >= N, "gemm_wtiled: ldr too small");
    GemmParams p;
    p.A = (const uint16_t*)A; p.W = (const uint16_t*)W_tiled; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual;
    p.C = (uint16_t*)C; p.C32 = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = K; p.ldc = ldc; p.ldr = ldr; p.act = act;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.splits = 1; p.kper = K / 64 + 1; p.part = nullptr; p.stages = 3; p.debug = 0; p.coal = 0;
    p.scale_m = p.scale_n = nullptr;
    p.w_tiled = 1;
    if (M > 64) return launch_gemm<128, 128>(p, 1, true, (hipStream_t)stream);
    return launch_gemm<64, 128>(p, 1, true, (hipStream_t)stream);
}
#endif   // FO1_ENABLE_AB

// fp8 linear (BASELINE configs[4], "fp8 MFMA"): C[M,N] = epilogue((Aq Wq^T) * scale_a[m] * scale_w[n]) with OCP e4m3 operands, fp32
// accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales), bf16 output; epilogues as fo1_gemm_bf16 (act 0..3).
// Aq [M, K] / Wq [N, K] bytes with lda / ldw in elements (= bytes); K % 128 == 0.  The reference has no fp8 path: the bar is the
// oracle's dequantised fp32 product (tests/test_fp8_gpu.py) and the tolerance table against the bf16 engine in DESIGN.md.
int fo1_gemm_fp8(const void* Aq, int lda, const float* scale_a, const void* Wq, int ldw, const float* scale_w, const void* bias,
                 const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* stream) {
    using namespace fo1;
    if (M == 0 || N == 0) return FO1_OK;
    FO1_CHECK_ARG(Aq && Wq && C && scale_a && scale_w, "gemm_fp8: NULL operand");
    FO1_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_fp8: bad shape M=%d N=%d K=%d", M, N, K);
    FO1_CHECK_ARG(K % 128 == 0 && lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K, "gemm_fp8: K %% 128, lda / ldw %% 16 (K=%d lda=%d ldw=%d)", K, lda, ldw);
    FO1_CHECK_ARG(((uintptr_t)Aq & 15) == 0 && ((uintptr_t)Wq & 15) == 0 && ((uintptr_t)scale_w & 15) == 0, "gemm_fp8: operands must be 16-byte aligned");
    FO1_CHECK_ARG(act >= 0 && act <= 3, "gemm_fp8: act=%d (0 none, 1 GELU, 2 SiLU, 3 interleaved SwiGLU)", act);
    FO1_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0, "gemm_fp8: N, ldc %% 4, C 8-byte aligned");
    FO1_CHECK_ARG((long long)M * lda < (1LL << 32) && (long long)N * ldw < (1LL << 32), "gemm_fp8: operands larger than 4 GB");
    FO1_CHECK_ARG(bias == nullptr || ((uintptr_t)bias & 7) == 0, "gemm_fp8: bias alignment");
    FO1_CHECK_ARG(residual == nullptr || (ldr % 4 == 0 && ((uintptr_t)residual & 7) == 0 && ldr >= N), "gemm_fp8: residual layout");
    if (act == 3) FO1_CHECK_ARG(residual == nullptr && N % 32 == 0 && ldc >= N / 2, "gemm_fp8: swiglu epilogue needs no residual, N %% 32 == 0, ldc >= N/2");
    else FO1_CHECK_ARG(ldc >= N, "gemm_fp8: ldc too small");
    GemmParams p;
    p.A = (const uint16_t*)Aq; p.W = (const uint16_t*)Wq; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual;
    p.C = (uint16_t*)C; p.C32 = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
    p.sA = p.sW = p.sC = p.sR = 0;
    p.splits = 1; p.kper = K / 128 + 1; p.part = nullptr; p.stages = 2; p.debug = 0; p.coal = 0;
    p.scale_m = scale_a; p.scale_n = scale_w;
    return launch_gemm_p4_fp8(p, (hipStream_t)stream);
}

// Row-wise e4m3 quantisation of a bf16 matrix: q[m, :] = e4m3(x[m, :] / scales[m]), scales[m] = absmax(x[m, :]) / 448.
int fo1_quantize_rows_e4m3(const void* x, long long ldx, int M, int K, void* q, long long ldq, float* scales, void* stream) {
    using namespace fo1;
    if (M == 0) return FO1_OK;
    FO1_CHECK_ARG(x && q && scales, "quantize_rows: NULL operand");
    FO1_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K, "quantize_rows: K, ldx, ldq %% 8 (K=%d)", K);
    FO1_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0, "quantize_rows: alignment");
    FO1_LAUNCH("quantize_rows_e4m3", (double)M * K * 3.0, quantize_rows_e4m3_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t*)x, ldx, K, (uint8_t*)q, ldq, scales);
    return FO1_OK;
}

}  // extern "C"
