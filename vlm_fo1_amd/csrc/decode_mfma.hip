// decode_mfma.hip — weight-streaming skinny GEMM for the decode step on gfx950: C[M <= 32, N] = epilogue(rmsnorm?(x) W^T) with
// the B sequences of a decode batch riding as the 16 columns of v_mfma_f32_16x16x32_bf16 (reference: the 1-token
// fast path of omchat_qwen2_5_vl.py:143-155 — q/k/v, o, gate/up, down, lm_head of modeling_qwen2_5_vl.py:731-734,633-635,1876).
//
// HBM-bound (every weight byte once, 6.2 GB per token): the kernel is a streaming engine, the matrix cores only keep the
// arithmetic off the VALU so that 8 or 16 sequences cost what one costs (the v_dot2 kernel of decode.hip spends 32 dot2 + 8
// LDS reads per 16 weight bytes at M = 8 and runs at 2.5 TB/s; here it is one MFMA per KB of weights).
//
//   * a workgroup (8 waves) owns one UNIT = NB blocks of 16 weight rows and walks K in steps of 64 elements; step s belongs
//     to wave s mod 8 (the K split is fixed by the shape alone, so a sequence's fp32 sum order does not depend on M: it
//     decodes to the same numbers alone and in any batch);
//   * weight loads keep the proven streaming shape — straight to VGPRs, non-temporal, one instruction = 8 rows x 128
//     contiguous bytes, 4 k-steps (up to 8 KB) in flight per wave, issued one K piece ahead and across unit boundaries —
//     and are turned into MFMA A fragments through a wave-private, XOR-swizzled 2 KB LDS scratch (conflict-free b128 both ways);
//   * x (M rows) lives in LDS with a (= 32 mod 256)-byte row pitch (conflict-free B-fragment reads), staged 2048 elements of
//     K at a time; Qwen2RMSNorm (modeling_qwen2_5_vl.py:126-140) is applied to the staged rows when asked;
//   * the 8 waves' partial accumulators meet in LDS (fixed order), wave 0 runs the epilogue: bias -> bf16 -> (+ residual) |
//     interleaved SwiGLU | QKV: bias -> bf16 -> mRoPE -> rotated q rows out, K row and V^T column appended to the caches.
//     Everything the epilogue reads from global memory (bias, residual, rope tables) is requested BEFORE the K loop.
//   * every row's sum is built the same way in every variant: a wave adds its k-steps s = wave + 8d into TWO chains (d even /
//     d odd), then chain A + chain B, then the 8 waves in a fixed tree.  That is what lets the HALF variant exist: for M <= 8 the
//     MFMA's 16 columns are only half used, so a unit can be 8 weight rows with the even-d k-step in fragment rows 0-7 /
//     columns 0-7 and the odd-d k-step in rows 8-15 / columns 8-15 (D's off-diagonal 8x8 blocks are ignored).  The few-row
//     projections (o, down: N = 2048 -> 128 units of 16 rows, half the CUs idle, 2.8 TB/s; q/k/v: 80 units) then fill the chip
//     with 256 / 160 units, without any cross-workgroup reduction and with bit-identical sums.
//   * R8 (round 3): the same 8-row units for 9..16 sequences.  All 16 MFMA columns carry sequences there, so the k-step pair (s, s + 8)
//     of a stage is two MFMAs: fragment rows 0-7 hold the 8 weight rows at k-step s (rows 8-15: the s + 8 data, their D rows are
//     ignored), then the fragment is read with rows swapped (fi ^ 8) for k-step s + 8.  Half of every MFMA is idle — the matrix cores
//     are not what a weight stream waits for — and o / down / q,k,v run on 256 / 256 / 160 workgroups instead of 128 / 128 / 80
//     (down at 16 sequences: 18.9 us = 2.4 TB/s before).  Chain A = first k-step of each pair, chain B = second: the same sets, the
//     same order, the same bits as the other variants.
//   * MM = 32 (round 3): 17..32 sequences as TWO column groups of the same MFMA per weight fragment — one stream of the weights for a whole
//     25-image pass instead of one per group of 16.  32 staged x rows of K = 2048 (132 KB) do not fit next to the weight scratch and the
//     reduction buffers, but a wave only ever multiplies by the k-steps s = wave + 8 d of x: for the single-piece projections (K <= 2048:
//     q/k/v, o, gate/up, lm_head) its B fragments are constants of the launch.  x is staged (and RMS-normalised) in LDS once, each wave
//     pulls its 4 k-steps x 2 halves x 2 groups = 16 fragments into 64 VGPRs, and the LDS region is then reused for scratch + reduction.
//     Deep-K projections (down) stage x in pieces of 16 k-steps (66 KB).  Waves 0 and 1 run the epilogue of group 0 / 1.  Sums per
//     (row, sequence) are built exactly as in the other variants.
#include <type_traits>

#include "decode_common.h"

namespace fo1 {

typedef __attribute__((ext_vector_type(8))) __bf16 gm_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gm_f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 gm_bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int gm_u32x4;

constexpr int GM_NW = 8;                           // waves per workgroup
constexpr int GM_NT = GM_NW * 64;
constexpr int GM_D = 4;                            // k-steps in flight per wave (register stages), full variant
constexpr int GM_PIECE = GM_NW * GM_D;             // k-steps of x staged at a time (32 x 64 = 2048 elements), full variant

__device__ __forceinline__ uint4 gm_load_nt16(const uint16_t* p) {
    const gm_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gm_u32x4*>(p));
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float gm_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float gm_dot8(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const gm_bf16x2*>(&a.x), *reinterpret_cast<const gm_bf16x2*>(&b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const gm_bf16x2*>(&a.y), *reinterpret_cast<const gm_bf16x2*>(&b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const gm_bf16x2*>(&a.z), *reinterpret_cast<const gm_bf16x2*>(&b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const gm_bf16x2*>(&a.w), *reinterpret_cast<const gm_bf16x2*>(&b.w), acc, false);
    return acc;
}
__device__ __forceinline__ void gm_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// MM: staged x rows (8 or 16).  NB: 16-row weight blocks per unit (SwiGLU / QKV pair two blocks whose rows meet in one lane).
// MP: K spans several staged pieces (deep-K projections); single-piece kernels stage x once and carry no reload logic.
// HALF: 8-row blocks, a register stage = the k-step pair (s, s + 8) of those rows (M <= 8 only; no SwiGLU form).
template <int MM, int MODE, int NB, bool MP, bool HALF, bool R8 = false>
__global__ __launch_bounds__(GM_NT) void gemv_mfma_kernel(const GemvBParams p, const int n_units, const int nsteps) {
    static_assert(MODE == GB_PLAIN || NB == 2, "paired modes use two row blocks");
    static_assert(!HALF || (MM == 8 && MODE != GB_SWIGLU), "HALF: M <= 8, plain or QKV");
    static_assert(!R8 || ((MM == 16 || MM == 32) && MODE != GB_SWIGLU && !HALF), "R8: 9..32 sequences, plain or QKV");
    static_assert(MM == 8 || MM == 16 || MM == 32, "8, 16 or 32 sequence columns");
    static_assert(MM != 32 || !HALF, "32 sequences: 16-row blocks, or 8-row blocks with the k-step pair as two MFMAs (R8, single-piece K only)");
    constexpr int NG = MM == 32 ? 2 : 1;                                 // column groups of 16 sequences
    constexpr int MR = MM == 32 ? 16 : MM;                               // sequences per group
    constexpr bool XREG = MM == 32 && !MP;                               // x fragments live in registers (single-piece K)
    constexpr bool H8 = HALF || R8;                                      // 8-row blocks, a stage = the k-step pair (s, s + 8)
    constexpr int PD = R8 ? 2 : ((HALF && !MP) ? 2 : ((MM == 32 && MP) ? 2 : 4));   // register stages per wave = stages per staged piece
    constexpr int SSTEP = H8 ? 16 : 8;                                   // k-step distance between a wave's consecutive stages
    constexpr int PSTEPS = SSTEP * PD;                                   // k-steps of x staged at a time: 32 (64: HALF && MP)
    constexpr int XPITCH = PSTEPS * 128 + 32;                            // bytes per staged x row (= 32 mod 256)
    constexpr int CPR = PSTEPS * 8;                                      // 16-byte chunks per staged x row
    constexpr int RPP = GM_NT / CPR;                                     // x rows per pass of the workgroup (2 or 1)
    constexpr int XL = MM / RPP;                                         // x loads per thread and piece
    constexpr int RB = H8 ? 8 : 16;                                      // weight rows per block
    extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
    unsigned char* const sx = gm_smem;                                   // [MM][XPITCH]
    // XR32 (round 6): 17..26 sequences, deep K (`down`), 8-row units — 32 staged rows of a 32-k-step piece (132 KB) do not fit beside the scratch and the
    // reduction buffers, M rows do up to M = 26: only the launch's own rows are staged, the column slots past them read the last staged row (never stored)
    constexpr bool XR32 = MM == 32 && MP && R8;
    const int xrows = XR32 ? p.M : MM;
    unsigned char* const sw = XREG ? gm_smem : gm_smem + xrows * XPITCH; // [GM_NW][NB][2048]  weight scratch (wave-private); XREG: over the dead x image
    float* const sred = reinterpret_cast<float*>(sw + GM_NW * NB * 2048);   // [2][NG][GM_NW][NB][64][4]
    float* const srstd = XREG ? reinterpret_cast<float*>(gm_smem + MM * XPITCH) : sred + 2 * NG * GM_NW * NB * 256;   // [32]
    static_assert(!XREG || GM_NW * NB * 2048 + 2 * NG * GM_NW * NB * 1024 <= MM * XPITCH, "scratch + reduction buffers fit in the x image");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kch = p.K >> 3;                                            // 16-byte chunks per row
    const int n_pieces = (nsteps + PSTEPS - 1) / PSTEPS;
    const int lrow = lane >> 3, lch = lane & 7;                          // load shape: row of an 8-row group, chunk of the 128-byte k-step
    const int fi = lane & 15, fg = lane >> 4;                            // fragment shape: MFMA row / column, k group
    unsigned char* const swv = sw + wave * (NB * 2048);
    const int xrow = MR == 16 ? fi : (fi & 7);
    const int n_rope = MODE == GB_QKV ? (p.n_q + p.n_kv) * (64 / RB) : 0;
    const int xsel = (HALF && fi >= 8) ? 8 * 128 : 0;                    // HALF: columns 8-15 take the pair's second k-step

    auto unit_rows = [&](int u, int (&rb)[NB]) __attribute__((always_inline)) {
        if (MODE == GB_SWIGLU) {
            rb[0] = u * 32;
            rb[NB - 1] = u * 32 + 16;
        } else if (MODE == GB_QKV) {
            constexpr int UPH = 64 / RB;                     // units per head: RB rotary-pair rows each
            if (u < n_rope) { rb[0] = (u / UPH) * 128 + (u % UPH) * RB; rb[NB - 1] = rb[0] + 64; }
            else { rb[0] = (p.n_q + p.n_kv) * 128 + (u - n_rope) * (2 * RB); rb[NB - 1] = rb[0] + RB; }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) rb[b] = (u * NB + b) * RB;
        }
    };
    auto unit_ptrs = [&](int u, const uint16_t* (&wp)[NB][2]) __attribute__((always_inline)) {
        int rb[NB];
        unit_rows(u < n_units ? u : n_units - 1, rb);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int row = rb[b] + (H8 ? 0 : q * 8) + lrow;    // 8-row blocks: both loads fetch the block's 8 rows (k-steps s and s + 8)
                row = row < p.N ? row : p.N - 1;            // clamp: the surplus rows' results are discarded
                wp[b][q] = p.W + (long long)row * p.ldw;
            }
    };
    // all four loads of one k-step of this wave; addresses are always valid (chunks past K are clamped: their x is zero in LDS)
    auto issue = [&](const uint16_t* const (&wp)[NB][2], int s, uint4 (&st)[NB][2]) __attribute__((always_inline)) {
        int c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            c[q] = (s + (H8 ? q * 8 : 0)) * 8 + lch;
            c[q] = c[q] < kch ? c[q] : kch - 1;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) st[b][q] = gm_load_nt16(wp[b][q] + (long long)c[q] * 8);
    };
    // acc: the chain this stage adds to (R8: chain A = the pair's first k-step; accb = chain B = its second)
    uint4 xf[XREG ? PD : 1][H8 ? 2 : 1][2][NG];           // XREG: B fragments of this wave's k-steps (stage d, k-step of the pair, half h, column group g)
    auto consume = [&](const uint4 (&st)[NB][2], int xs, int d, gm_f32x4 (&acc)[NG][NB], gm_f32x4 (&accb)[NG][NB]) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rl = q * 8 + lrow;
                *reinterpret_cast<uint4*>(swv + b * 2048 + rl * 128 + ((lch ^ ((rl >> 1) & 7)) << 4)) = st[b][q];
            }
        gm_lds_fence();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 xv[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if constexpr (XREG) xv[g] = xf[d][0][h][g];
                else xv[g] = *reinterpret_cast<const uint4*>(sx + (XR32 ? min(g * 16 + xrow, xrows - 1) : g * 16 + xrow) * XPITCH + xs * 128 + xsel + h * 64 + fg * 16);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint4 wv = *reinterpret_cast<const uint4*>(swv + b * 2048 + fi * 128 + (((h * 4 + fg) ^ ((fi >> 1) & 7)) << 4));
#pragma unroll
                for (int g = 0; g < NG; ++g)      // one weight fragment, NG column groups of 16 sequences
                    acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<gm_bf16x8*>(&wv), *reinterpret_cast<gm_bf16x8*>(&xv[g]), acc[g][b], 0, 0, 0);
            }
            if (R8) {     // the pair's second k-step: scratch rows 8-15 read as fragment rows 0-7 (fi ^ 8), x one k-step-pair half further
                const int fj = fi ^ 8;
                uint4 xw[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if constexpr (XREG) xw[g] = xf[d][H8 ? 1 : 0][h][g];
                    else xw[g] = *reinterpret_cast<const uint4*>(sx + (XR32 ? min(g * 16 + xrow, xrows - 1) : g * 16 + xrow) * XPITCH + (xs + 8) * 128 + h * 64 + fg * 16);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    uint4 wv = *reinterpret_cast<const uint4*>(swv + b * 2048 + fj * 128 + (((h * 4 + fg) ^ ((fj >> 1) & 7)) << 4));
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        accb[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<gm_bf16x8*>(&wv), *reinterpret_cast<gm_bf16x8*>(&xw[g]), accb[g][b], 0, 0, 0);
                }
            }
        }
        gm_lds_fence();      // the scratch is rewritten by the next k-step
    };
    // x rows of K piece `piece`: global -> registers (clamped addresses, no branch), then registers -> LDS (zero beyond M rows /
    // beyond K) and the fused RMSNorm (single-piece K only: host-checked).  Split in two so that the loads can be issued BEFORE the
    // weight refills of the piece in progress: vmcnt retires in order, a younger x load would drain the whole weight pipeline.
    auto load_x = [&](int piece, uint4 (&t)[XL]) __attribute__((always_inline)) {
        const int c = tid % CPR, gc = piece * CPR + c;
        const int gcc = gc < kch ? gc : kch - 1;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int m = tid / CPR + RPP * i;
            t[i] = *reinterpret_cast<const uint4*>(p.X + (long long)(m < p.M ? m : p.M - 1) * p.ldx + gcc * 8);
        }
    };
    auto store_x = [&](int piece, const uint4 (&t)[XL], const uint4& nw) __attribute__((always_inline)) {
        const int c = tid % CPR, gc = piece * CPR + c;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int m = tid / CPR + RPP * i;
            const bool ok = m < p.M && gc < kch;
            if (!XR32 || m < xrows) *reinterpret_cast<uint4*>(sx + m * XPITCH + c * 16) = ok ? t[i] : uint4{0, 0, 0, 0};
        }
        __syncthreads();
        if (!MP && p.norm_w) {
            for (int m = wave; m < MM; m += GM_NW) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 v = *reinterpret_cast<const uint4*>(sx + m * XPITCH + (lane + 64 * i) * 16);
                    ss = gm_dot8(v, v, ss);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                if (lane == 0) srstd[m] = rsqrtf(ss / (float)p.K + p.norm_eps);
            }
            __syncthreads();
            // fp32 variance, bf16(x * rstd), * weight -> bf16 (the reference's rounding points)
#pragma unroll
            for (int i = 0; i < XL; ++i) {             // (!MP: 256 chunks per row, two rows per pass)
                const int m = tid / CPR + RPP * i;
                const float rstd = srstd[m];
                uint4* px = reinterpret_cast<uint4*>(sx + m * XPITCH + c * 16);
                const uint4 v = *px;
                uint4 o;
                o.x = pack_bf16x2(bf16_lo(nw.x) * gm_round(bf16_lo(v.x) * rstd), bf16_hi(nw.x) * gm_round(bf16_hi(v.x) * rstd));
                o.y = pack_bf16x2(bf16_lo(nw.y) * gm_round(bf16_lo(v.y) * rstd), bf16_hi(nw.y) * gm_round(bf16_hi(v.y) * rstd));
                o.z = pack_bf16x2(bf16_lo(nw.z) * gm_round(bf16_lo(v.z) * rstd), bf16_hi(nw.z) * gm_round(bf16_hi(v.z) * rstd));
                o.w = pack_bf16x2(bf16_lo(nw.w) * gm_round(bf16_lo(v.w) * rstd), bf16_hi(nw.w) * gm_round(bf16_hi(v.w) * rstd));
                *px = o;
            }
            __syncthreads();
        }
    };

    // ---- prologue: the weight stream does not depend on x — the first unit's first piece is requested before anything else.
    // NO branch encloses a global load from here on: hipcc turns a conditionally issued load into "load, then s_waitcnt vmcnt(0) at
    // the join" (measured: the first form of this loop ran every k-step at full HBM latency, 3.4 TB/s); addresses are clamped to
    // valid memory instead and the surplus data meets zeros (x beyond K) or is never stored (rows beyond N). ----
    // x (tiny, L2-resident, needed first) is requested before the weights: waiting for it then leaves the weight loads in flight.
    const uint16_t* const dummy = p.W;                       // any valid address for the operands a launch does not have
    uint4 nw;
    {
        const int c = tid & 255;
        nw = *reinterpret_cast<const uint4*>((p.norm_w ? p.norm_w : dummy) + (c < kch ? c : kch - 1) * 8);
    }
    uint4 xr[XL];
    load_x(0, xr);
    uint4 st[PD][NB][2];
    const uint16_t* wcur[NB][2];
    unit_ptrs(blockIdx.x, wcur);
#pragma unroll
    for (int d = 0; d < PD; ++d) issue(wcur, wave + SSTEP * d, st[d]);
    // decode state of this lane's sequence (QKV epilogue): cache row and rope-table row.  Wave g < NG runs the epilogue of column group g.
    const int eg = wave < NG ? wave : 0;
    const int n_seq = eg * 16 + fi;
    const bool seq_ok = n_seq < p.M && (!H8 || fg < 2);          // 8-row blocks: D rows 8-15 are chain B (HALF, folded into lanes fg < 2) or unused (R8)
    int pos = 0;
    long long trow = 0;
    if (MODE == GB_QKV) {
        const int* stt = p.state + (seq_ok ? n_seq : 0) * 8;
        pos = stt[0];
        trow = stt[1];
    }
    if constexpr (!MP && MM == 8) {
        if (p.attn_part != nullptr) {
            // x rows = the attention output, combined from the split-KV partials right here (M <= 2: one (row, 16-byte chunk) per thread): what
            // attn_decode_combine_kernel + the plain x staging produce, bit for bit (attn_combine_row is the arithmetic of both)
            __syncthreads();
            const int c = tid % CPR, m0 = tid / CPR;               // CPR = 256 chunks of K = 2048: rows 0, 1
            uint4 o = uint4{0, 0, 0, 0};
            if (m0 < p.M && c < kch) {
                const int* stt = p.attn_state + m0 * 8;
                if (!stt[3]) {
                    const int head = c >> 4, d0 = (c & 15) * 8;
                    const int kvh = head / p.attn_group, slot = head - kvh * p.attn_group;
                    const int n_valid = (stt[0] + 1 - stt[2] + p.attn_chunk - 1) / p.attn_chunk;
                    float v[8];
                    attn_combine_row<8>(p.attn_part + (long long)m0 * p.attn_part_seq_stride + ((long long)kvh * 16 + slot) * 130,
                                        (long long)p.attn_n_kv * 16 * 130, n_valid, d0, v);
                    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                }
            }
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int m = m0 + RPP * i;
                *reinterpret_cast<uint4*>(sx + m * XPITCH + c * 16) = i == 0 ? o : uint4{0, 0, 0, 0};
            }
            __syncthreads();
        } else {
            store_x(0, xr, nw);
        }
    } else if (!MP) store_x(0, xr, nw);      // single piece: x staged (and normalised) once for every unit of this workgroup
    if constexpr (XREG) {
        // this wave's B fragments — k-steps wave + 8 d, both halves, both column groups — are the same for every unit: into registers,
        // then the x image is dead and its LDS becomes the weight scratch and the reduction buffers
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int q = 0; q < (H8 ? 2 : 1); ++q)          // 8-row blocks: a stage is the k-step pair (s, s + 8)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        xf[d][q][h][g] = *reinterpret_cast<const uint4*>(sx + (g * 16 + fi) * XPITCH + (wave + SSTEP * d + 8 * q) * 128 + h * 64 + fg * 16);
        __syncthreads();
    }
    const uint16_t* const bias_src = p.bias ? p.bias : dummy;
    const uint16_t* const res_src = p.res ? p.res : dummy;
    const long long res_ld = p.res ? p.ldr : 0;

    // One unit.  PF (compile time): the workgroup has another unit after this one — its first piece is requested while this
    // unit's last piece is consumed.  The caller peels the last unit, so no load sits under a run-time condition.
    auto unit_body = [&](int u, int it, auto pf_tag) __attribute__((always_inline)) {
        constexpr bool PF = decltype(pf_tag)::value;
        const uint16_t* wnext[NB][2];
        if (PF) unit_ptrs(u + gridDim.x, wnext);
        int rb[NB];
        unit_rows(u, rb);
        // ---- epilogue operands, requested now (every wave: no branch), used after the K loop by wave 0 ----
        uint2 e_bias[NB], e_res[NB], e_cos[2], e_sin[2];    // 4 consecutive bf16 each (N % 4 == 0, ldr % 4 == 0: host-checked)
        {
            const int nc = seq_ok ? n_seq : 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int f0 = rb[b] + fg * 4;
                f0 = f0 + 4 <= p.N ? f0 : p.N - 4;
                e_bias[b] = *reinterpret_cast<const uint2*>(bias_src + f0);
                e_res[b] = MODE == GB_PLAIN ? *reinterpret_cast<const uint2*>(res_src + (long long)nc * res_ld + f0) : uint2{0, 0};
                if (!p.bias) e_bias[b] = uint2{0, 0};
                if (!p.res) e_res[b] = uint2{0, 0};
            }
            if (MODE == GB_QKV) {
                const int d0 = ((u % (64 / RB)) * RB + fg * 4) & 63;        // (V units / HALF's unused lanes load a harmless table row too)
                e_cos[0] = *reinterpret_cast<const uint2*>(p.cos_t + trow * 128 + d0);
                e_sin[0] = *reinterpret_cast<const uint2*>(p.sin_t + trow * 128 + d0);
                e_cos[1] = *reinterpret_cast<const uint2*>(p.cos_t + trow * 128 + d0 + 64);
                e_sin[1] = *reinterpret_cast<const uint2*>(p.sin_t + trow * 128 + d0 + 64);
            } else {
                e_cos[0] = e_cos[1] = e_sin[0] = e_sin[1] = uint2{0, 0};
            }
        }
        // two chains per row: k-steps wave + 8d with d even / d odd (full: stages alternate; HALF: rows 0-7 / 8-15 of one stage)
        gm_f32x4 accs[2][NG][NB];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int b = 0; b < NB; ++b) accs[0][g][b] = accs[1][g][b] = gm_f32x4{0.f, 0.f, 0.f, 0.f};
        gm_f32x4 (&acc)[NG][NB] = accs[0];
        gm_f32x4 (&acc2)[NG][NB] = accs[1];

        // every piece but the last: consume a k-step, refill its stage with the k-step one piece ahead in the same unit
        if (MP) {
            for (int piece = 0; piece + 1 < n_pieces; ++piece) {
                __syncthreads();                 // every wave is done with the previous piece of x
                store_x(piece, xr, nw);
                load_x(piece + 1, xr);           // before this piece's weight refills (in-order vmcnt)
#pragma unroll
                for (int d = 0; d < PD; ++d) {
                    const int xs = wave + SSTEP * d;              // k-step inside the staged piece
                    consume(st[d], xs, d, accs[H8 ? 0 : (d & 1)], accs[1]);
                    issue(wcur, (piece + 1) * PSTEPS + xs, st[d]);
                }
            }
            __syncthreads();
            store_x(n_pieces - 1, xr, nw);
            load_x(0, xr);                       // the next unit's first piece (unused after the workgroup's last unit)
        }
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int xs = wave + SSTEP * d;
            consume(st[d], xs, d, accs[H8 ? 0 : (d & 1)], accs[1]);
            if (PF) issue(wnext, xs, st[d]);                  // the first piece of this workgroup's next unit
        }
        // chain A + chain B.  HALF: chain B of (row r, sequence n) sits in lane + 40 (fragment row r + 8, column n + 8)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (HALF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc2[g][b][r] = __shfl(acc[g][b][r], (lane + 40) & 63, 64);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][b][r] += acc2[g][b][r];
            }
        // ---- the 8 waves' partial sums meet in LDS; double-buffered by unit parity: one barrier per unit ----
        float* red = sred + (it & 1) * (NG * GM_NW * NB * 256);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                *reinterpret_cast<float4*>(red + (((g * GM_NW + wave) * NB + b) * 64 + lane) * 4) =
                    float4{acc[g][b][0], acc[g][b][1], acc[g][b][2], acc[g][b][3]};
        __syncthreads();
        if (wave < NG) {        // wave g finishes column group g (sequences 16 g .. 16 g + 15)
            auto h4 = [](const uint2& q, int r) -> float { const uint32_t w = (r >> 1) ? q.y : q.x; return (r & 1) ? bf16_hi(w) : bf16_lo(w); };
            float v[NB][4];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float4 t[GM_NW];
#pragma unroll
                for (int w = 0; w < GM_NW; ++w) t[w] = *reinterpret_cast<const float4*>(red + (((eg * GM_NW + w) * NB + b) * 64 + lane) * 4);
                v[b][0] = ((t[0].x + t[1].x) + (t[2].x + t[3].x)) + ((t[4].x + t[5].x) + (t[6].x + t[7].x));
                v[b][1] = ((t[0].y + t[1].y) + (t[2].y + t[3].y)) + ((t[4].y + t[5].y) + (t[6].y + t[7].y));
                v[b][2] = ((t[0].z + t[1].z) + (t[2].z + t[3].z)) + ((t[4].z + t[5].z) + (t[6].z + t[7].z));
                v[b][3] = ((t[0].w + t[1].w) + (t[2].w + t[3].w)) + ((t[4].w + t[5].w) + (t[6].w + t[7].w));
            }
            // v[b][r] = sum_k W[rb[b] + fg*4 + r][k] x[n_seq][k]
            if (seq_ok) {
                if (MODE == GB_PLAIN) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int f0 = rb[b] + fg * 4;
                        uint16_t o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = v[b][r] + h4(e_bias[b], r);
                            x = gm_round(x);
                            x += h4(e_res[b], r);
                            o[r] = f32_to_bf16(x);
                        }
                        uint16_t* cp = p.C + (long long)n_seq * p.ldc + f0;
                        if (f0 < p.N) {
                            if ((p.ldc & 3) == 0) {
                                *reinterpret_cast<uint2*>(cp) = uint2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) cp[r] = o[r];
                            }
                        }
                    }
                } else if (MODE == GB_SWIGLU) {
                    // rows [32u, 32u+16) = gate of features 16u.., rows [32u+16, 32u+32) = their up partners
                    uint16_t o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float g = v[0][r] + h4(e_bias[0], r), up = v[NB - 1][r] + h4(e_bias[NB - 1], r);
                        g = gm_round(g);
                        up = gm_round(up);
                        o[r] = f32_to_bf16(gm_round(fo1_silu(g)) * up);
                    }
                    uint16_t* cp = p.C + (long long)n_seq * p.ldc + u * 16 + fg * 4;
                    if ((p.ldc & 3) == 0) {
                        *reinterpret_cast<uint2*>(cp) = uint2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) cp[r] = o[r];
                    }
                } else {   // GB_QKV
                    if (u < n_rope) {
                        const int head = u / (64 / RB), d0 = (u % (64 / RB)) * RB + fg * 4;        // d0 + r < 64, rotary partner d + 64
                        uint16_t oa[4], ob[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float a = v[0][r] + h4(e_bias[0], r), b = v[NB - 1][r] + h4(e_bias[NB - 1], r);
                            a = gm_round(a);                                          // the bf16 q/k the unfused path stores
                            b = gm_round(b);
                            const float c1 = h4(e_cos[0], r), s1 = h4(e_sin[0], r), c2 = h4(e_cos[1], r), s2 = h4(e_sin[1], r);
                            oa[r] = f32_to_bf16(gm_round(a * c1) + gm_round(-b * s1));     // rotate_half, three bf16 roundings
                            ob[r] = f32_to_bf16(gm_round(b * c2) + gm_round(a * s2));
                        }
                        const uint2 pa = uint2{(uint32_t)oa[0] | ((uint32_t)oa[1] << 16), (uint32_t)oa[2] | ((uint32_t)oa[3] << 16)};
                        const uint2 pb = uint2{(uint32_t)ob[0] | ((uint32_t)ob[1] << 16), (uint32_t)ob[2] | ((uint32_t)ob[3] << 16)};
                        if (head < p.n_q) {
                            uint16_t* qp = p.C + (long long)n_seq * p.ldc + head * 128 + d0;
                            if ((p.ldc & 3) == 0) {
                                *reinterpret_cast<uint2*>(qp) = pa;
                                *reinterpret_cast<uint2*>(qp + 64) = pb;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { qp[r] = oa[r]; qp[64 + r] = ob[r]; }
                            }
                        } else {
                            uint16_t* kc = p.kcache + (long long)(head - p.n_q) * p.kc_head_stride + (long long)pos * 128 + d0;
                            *reinterpret_cast<uint2*>(kc) = pa;
                            *reinterpret_cast<uint2*>(kc + 64) = pb;
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int vrow = (u - n_rope) * (2 * RB) + b * RB + fg * 4 + r;          // kv_head * 128 + d
                                p.vtcache[(long long)vrow * p.vt_row_stride + pos] = f32_to_bf16(v[b][r] + h4(e_bias[b], r));
                            }
                    }
                }
            }
        }
        if (PF) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) wcur[b][q] = wnext[b][q];
        }
    };

    int it = 0, u = blockIdx.x;
    for (; u + (int)gridDim.x < n_units; u += gridDim.x, ++it) unit_body(u, it, std::true_type{});
    unit_body(u, it, std::false_type{});      // the workgroup's last unit (grid <= n_units: every workgroup has one)
}

extern int g_gemv_profile_shapes;

template <int MM, int MODE, int NB, bool MP, bool HALF, bool R8 = false>
static int launch_gemv_mfma_mp(const GemvBParams& p, int n_units, int nsteps, const char* name, hipStream_t st) {
    constexpr int pd = R8 ? 2 : ((HALF && !MP) ? 2 : ((MM == 32 && MP) ? 2 : 4));
    constexpr int xpitch = ((HALF || R8) ? 16 : 8) * pd * 128 + 32;
    constexpr int ng = MM == 32 ? 2 : 1;
    // MM == 32, single piece: the weight scratch and the reduction buffers reuse the x image once its fragments sit in registers
    const size_t xrows = (MM == 32 && MP && R8) ? (size_t)p.M : (size_t)MM;       // (deep-K 8-row units at 17..26 sequences: the launch's own rows only)
    const size_t smem = (MM == 32 && !MP) ? (size_t)MM * xpitch + 128
                                          : xrows * xpitch + (size_t)GM_NW * NB * 2048 + (size_t)2 * ng * GM_NW * NB * 1024 + 128;
    if (smem > 156 * 1024) return set_err(FO1_ERR_ARG, "gemv_batch: %zu B of LDS for M=%d K=%d (internal dispatch error)", smem, p.M, p.K);
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_mfma_kernel<MM, MODE, NB, MP, HALF, R8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        attr = true;
    }
    // persistent workgroups, one per CU (x is staged / normalised once per workgroup when K fits one piece)
    const int grid = n_units < 256 ? n_units : 256;
    FO1_LAUNCH(name, (double)p.N * p.K * 2.0, (gemv_mfma_kernel<MM, MODE, NB, MP, HALF, R8>), dim3(grid), dim3(GM_NT), smem, st, p, n_units, nsteps);
    return FO1_OK;
}

template <int MM, int MODE, int NB, bool HALF = false, bool R8 = false>
static int launch_gemv_mfma(const GemvBParams& p, int n_units, int nsteps, const char* name, hipStream_t st) {
    constexpr int piece = R8 ? 32 : GM_PIECE;        // k-steps of x staged at a time (R8 stages 2 pairs per wave); 32 sequences: one piece up to K = 2048, pieces of 16 beyond
    if (nsteps > piece) return launch_gemv_mfma_mp<MM, MODE, NB, true, HALF, R8>(p, n_units, nsteps, name, st);
    return launch_gemv_mfma_mp<MM, MODE, NB, false, HALF, R8>(p, n_units, nsteps, name, st);
}

#ifdef FO1_ENABLE_AB
int g_gemv_half = 3;     // bit 0: M <= 8 8-row units (HALF); bit 1: 9..32 sequences 8-row units (R8; 17..32: single-piece K only) — few-row projections (A/B: fo1_gemv_batch_set_impl)
#endif

template <int MM>
static int dispatch_gemv_mfma(GemvBParams& p, int mode, hipStream_t st) {
    const int nsteps = cdiv(p.K, 64);
    if (p.norm_w && nsteps > GM_PIECE) return set_err(FO1_ERR_ARG, "gemv_batch: fused RMSNorm needs K <= %d (K=%d)", GM_PIECE * 64, p.K);
    // 17..32 sequences with a deep K (x staged in pieces) and a paired mode (two row blocks per unit: QKV, SwiGLU, 32-row plain units)
    // would need 32 * 2080 + 8 * 2 * 2048 + 2 * 2 * 8 * 2 * 1024 + 128 = 164 992 B of LDS — more than the 156 KB the launch may ask
    // for.  The shipped model (hidden 2048) never gets here; refuse instead of failing the launch.
    if (MM == 32 && nsteps > 32 && (mode != GB_PLAIN || p.N >= 8192))
        return set_err(FO1_ERR_ARG, "gemv_batch: M > 16 with K > 2048 is built for the plain few-row form only (M=%d N=%d K=%d mode=%d)", p.M, p.N, p.K, mode);
    char pname[56];
    const char* name = mode == GB_SWIGLU ? "gemv_mfma_swiglu" : (mode == GB_QKV ? "gemv_mfma_qkv" : "gemv_mfma");
    if (profile_enabled() && g_gemv_profile_shapes) {
        snprintf(pname, sizeof pname, "gemv_mfma m%d %dx%d mode%d", p.M, p.N, p.K, mode);
        name = pname;
    }
    if (mode == GB_SWIGLU) return launch_gemv_mfma<MM, GB_SWIGLU, 2>(p, p.N / 32, nsteps, name, st);
    if constexpr (MM == 8) {
        // M <= 8: half of the MFMA's columns are free — 8-row units with the k-step pair in the two halves (same sums, see the header)
        if ((g_gemv_half & 1) && mode == GB_QKV) return launch_gemv_mfma<8, GB_QKV, 2, true>(p, (p.n_q + p.n_kv) * 8 + p.n_kv * 8, nsteps, name, st);
        if ((g_gemv_half & 1) && mode == GB_PLAIN && p.N <= 4096) return launch_gemv_mfma<8, GB_PLAIN, 1, true>(p, cdiv(p.N, 8), nsteps, name, st);
    }
    if constexpr (MM == 16) {
        // 9..16 sequences: 8-row units with both k-steps of a pair as separate MFMAs (R8, see the header) — the few-row projections
        // fill the chip like they do at M <= 8; bit 2 of fo1_gemv_batch_set_impl's half switch (g_gemv_half & 2 == 0) turns it off
        if ((g_gemv_half & 2) && mode == GB_QKV) return launch_gemv_mfma<16, GB_QKV, 2, false, true>(p, (p.n_q + p.n_kv) * 8 + p.n_kv * 8, nsteps, name, st);
        if ((g_gemv_half & 2) && mode == GB_PLAIN && p.N <= 4096) return launch_gemv_mfma<16, GB_PLAIN, 1, false, true>(p, cdiv(p.N, 8), nsteps, name, st);
    }
    if constexpr (MM == 32) {
        // 17..32 sequences, single-piece K only (a deep-K piece of 32 x 4 KB rows does not fit next to the buffers): the same 8-row units
        if ((g_gemv_half & 2) && nsteps <= 32 && mode == GB_QKV) return launch_gemv_mfma_mp<32, GB_QKV, 2, false, false, true>(p, (p.n_q + p.n_kv) * 8 + p.n_kv * 8, nsteps, name, st);
        if ((g_gemv_half & 2) && nsteps <= 32 && mode == GB_PLAIN && p.N <= 4096) return launch_gemv_mfma_mp<32, GB_PLAIN, 1, false, false, true>(p, cdiv(p.N, 8), nsteps, name, st);
        // deep K (`down`), 17..27 sequences: the same 8-row units with x staged in 32-k-step pieces of the launch's OWN rows (M x 4 KB + 48 KB of LDS fit up to
        // M = 26) — 256 workgroups where the 16-row units below have 128, the same sums; 27..32 sequences keep the 16-row units (bit 2 of the half switch off: A/B)
        if ((g_gemv_half & 2) && nsteps > 32 && mode == GB_PLAIN && p.N <= 4096 && p.M <= 26 && !p.norm_w)
            return launch_gemv_mfma_mp<32, GB_PLAIN, 1, true, false, true>(p, cdiv(p.N, 8), nsteps, name, st);
    }
    if (mode == GB_QKV) return launch_gemv_mfma<MM, GB_QKV, 2>(p, (p.n_q + p.n_kv) * 4 + p.n_kv * 4, nsteps, name, st);
    // plain: 16-row units for the few-row projections (every CU should stream), 32-row units for lm_head-sized matrices
    if (p.N >= 8192) return launch_gemv_mfma<MM, GB_PLAIN, 2>(p, cdiv(p.N, 32), nsteps, name, st);
    return launch_gemv_mfma<MM, GB_PLAIN, 1>(p, cdiv(p.N, 16), nsteps, name, st);
}

int gemv_mfma_any(GemvBParams& p, int mode, hipStream_t st) {
    if (p.M <= 8) return dispatch_gemv_mfma<8>(p, mode, st);
    if (p.M <= 16) return dispatch_gemv_mfma<16>(p, mode, st);
    return dispatch_gemv_mfma<32>(p, mode, st);
}

}  // namespace fo1
