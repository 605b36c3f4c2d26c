// decode.hip — batched greedy decode for gfx950: B sequences advance one token per step through ONE stream of the weights
// (reference: the 1-token fast path of omchat_qwen2_5_vl.py:143-155, positions modeling_qwen2_5_vl.py:1848-1860,
// stop rule mm_utils.py:137-181 / HF greedy search).  Everything position-dependent lives in device memory so that one
// captured hipGraph serves every step of every batch:
//
//   per-sequence state  int32[8] = { pos, rope_row, kv_start, finished, n_gen, max_new, -, - }
//     pos       cache row the NEW token's K / V^T column is written to (= kv_start + tokens so far)
//     rope_row  row of the [max positions, head_dim] mRoPE table = cache_position + rope_delta (text: t = h = w)
//     kv_start  first cache row of the sequence's slot; keys attended = [kv_start, pos]
//
// Kernels (5 launches per layer):
//   gemv_batch<MM, MODE>   weight-streaming GEMV for M <= 8 rows: 16-B weight loads straight to VGPRs, x in LDS,
//                          v_dot2c_f32_bf16, fp32 accumulate; fused RMSNorm prologue; epilogues: bias/residual | interleaved
//                          SwiGLU | QKV = bias -> bf16 -> mRoPE -> q rows out, K row and V^T column appended to the caches
//   attention              split-KV partials (attention.hip, PARTIAL mode with per-sequence ranges) + fixed-order combine
//   argmax_accept          two-stage argmax per row, then ON-DEVICE bookkeeping: record the id, stop check, advance state,
//                          write the next step's embedding-gather plan — the host never reads a token inside the loop
#include "decode_common.h"
#include "ab.h"

namespace fo1 {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// streamed-once weights: non-temporal 16-byte load (MI355X_MICROARCH "nt-weights": -18 % issue-to-landed on a decode weight stream)
__device__ __forceinline__ uint4 load_nt16(const uint16_t* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return uint4{v.x, v.y, v.z, v.w};
}

__device__ __forceinline__ float gb_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float gb_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dot8b(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.x), *reinterpret_cast<const bf16x2_t*>(&x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.y), *reinterpret_cast<const bf16x2_t*>(&x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.z), *reinterpret_cast<const bf16x2_t*>(&x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.w), *reinterpret_cast<const bf16x2_t*>(&x.w), acc, false);
    return acc;
}

// One unit = 8 weight rows:  plain: 8 consecutive output features;  SwiGLU: 4 gate rows + their up partners 16 rows further
// (16-row interleaved weights);  QKV: q/k heads -> 4 dims d and their rotary partners d + 64, v head -> 8 consecutive dims.
// Lane mapping: lane l streams row (l >> 3) of the unit, 16-B chunk (l & 7) + 8 i of K — a load instruction covers 8 rows x 128
// contiguous bytes — and keeps only MM accumulators; a row's 8 lanes are reduced with three DPP-friendly xor shuffles (a
// lane-per-chunk mapping would need 6 shuffles for each of 8 x MM sums: 384 ds_bpermute per unit at MM = 8).
// KSPLIT: the 4 waves of a workgroup share a unit and split K (few-row projections: every CU streams); otherwise one unit per
// wave.  Workgroups are persistent over units (grid-stride): x is staged (and RMS-normalised) once per workgroup.
// RPL (rows per lane): a lane streams row j of RPL consecutive units at the same chunk position and shares every x chunk it reads
// from LDS between them.  At M = 8 one unit per lane reads 8 x-chunks per weight chunk — LDS traffic 8x the weight stream, 9 us
// for the 90 MB gate/up matrix on its own; RPL = 4 brings it to 2x.  The per-(row, sequence) sum order does not depend on RPL
// (or on M): a sequence decodes to the same numbers alone and in any batch.
#ifdef FO1_ENABLE_AB      // the v_dot2 streaming kernel: equal at 1 sequence, 1.4x slower at 8 than the MFMA skinny GEMM; A/B only
template <int MM, int MODE, bool KSPLIT, int RPL>
__global__ __launch_bounds__(256) void gemv_batch_kernel(const GemvBParams p) {
    constexpr int NR = 8, U = 8 / RPL;   // 16-B weight loads per row per batch; two batches (register buffers) in flight per lane
    extern __shared__ __attribute__((aligned(16))) uint16_t sx[];   // [MM][kp_chunks * 8]
    __shared__ float s_red[4][RPL * NR * MM];
    __shared__ float s_rstd[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jrow = lane >> 3, kslot = lane & 7;
    const int kch = p.K >> 3;
    const int n_rope = (MODE == GB_QKV) ? (p.n_q + p.n_kv) * 16 : 0;
    int n_units;
    if (MODE == GB_SWIGLU) n_units = p.N / 8;            // 4 features per unit
    else if (MODE == GB_QKV) n_units = n_rope + p.n_kv * 16;
    else n_units = (p.N + 7) / 8;
    const bool single_piece = p.kp_chunks >= kch;

    auto stage_x = [&](int kp0, int kpn) {
        // thread t owns chunk columns c = t, t + 256, ... of ALL MM rows: MM independent 16-B loads in flight per batch (a
        // row-major sweep is MM * K / 2048 dependent L2 round trips per thread: 8 x ~0.7 us at M = 8, K = 2048)
        for (int c = tid; c < kpn; c += 256) {
            uint4 t[MM];
#pragma unroll
            for (int m = 0; m < MM; ++m)
                t[m] = m < p.M ? *reinterpret_cast<const uint4*>(p.X + (long long)m * p.ldx + (kp0 + c) * 8) : uint4{0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MM; ++m) *reinterpret_cast<uint4*>(&sx[(m * p.kp_chunks + c) * 8]) = t[m];
        }
        __syncthreads();
        if (p.norm_w) {
            // fused Qwen2RMSNorm (modeling_qwen2_5_vl.py:126-140) on the staged rows (K fits one piece: checked by the host):
            // fp32 variance, bf16(x * rstd), * weight -> bf16.  Wave w reduces rows w, w + 4: no workgroup barrier per row.
            for (int m = wave; m < MM; m += 4) {
                float ss = 0.f;
                for (int c = lane; c < kpn; c += 64) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&sx[(m * p.kp_chunks + c) * 8]);
                    ss = dot8b(v, v, ss);
                }
                ss = gb_wave_sum(ss);
                if (lane == 0) s_rstd[m] = rsqrtf(ss / (float)p.K + p.norm_eps);
            }
            __syncthreads();
            for (int c = tid; c < kpn; c += 256) {       // one norm-weight load per chunk column, then LDS only
                const uint4 w = *reinterpret_cast<const uint4*>(p.norm_w + c * 8);
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const float rstd = s_rstd[m];
                    const uint4 v = *reinterpret_cast<const uint4*>(&sx[(m * p.kp_chunks + c) * 8]);
                    uint4 o;
                    o.x = pack_bf16x2(bf16_lo(w.x) * gb_round(bf16_lo(v.x) * rstd), bf16_hi(w.x) * gb_round(bf16_hi(v.x) * rstd));
                    o.y = pack_bf16x2(bf16_lo(w.y) * gb_round(bf16_lo(v.y) * rstd), bf16_hi(w.y) * gb_round(bf16_hi(v.y) * rstd));
                    o.z = pack_bf16x2(bf16_lo(w.z) * gb_round(bf16_lo(v.z) * rstd), bf16_hi(w.z) * gb_round(bf16_hi(v.z) * rstd));
                    o.w = pack_bf16x2(bf16_lo(w.w) * gb_round(bf16_lo(v.w) * rstd), bf16_hi(w.w) * gb_round(bf16_hi(v.w) * rstd));
                    *reinterpret_cast<uint4*>(&sx[(m * p.kp_chunks + c) * 8]) = o;
                }
            }
            __syncthreads();
        }
    };
    // this lane's weight row of a unit
    auto unit_row = [&](int unit) -> const uint16_t* {
        const int u = unit < n_units ? unit : n_units - 1, j = jrow;
        int row;
        if (MODE == GB_SWIGLU) { const int f = u * 4 + (j & 3); row = (f >> 4) * 32 + (f & 15) + (j >= 4 ? 16 : 0); }
        else if (MODE == GB_QKV) {
            if (u < n_rope) row = (u >> 4) * 128 + (u & 15) * 4 + (j & 3) + (j >= 4 ? 64 : 0);
            else row = (p.n_q + p.n_kv) * 128 + (u - n_rope) * 8 + j;
        } else row = u * 8 + j;
        row = row < p.N ? row : p.N - 1;     // clamp (result discarded)
        return p.W + (long long)row * p.ldw;
    };
    // this wave's chunk range inside a K piece of kpn chunks (multiples of 8: one chunk per lane of a row group)
    auto wave_range = [&](int kpn, int& c_begin, int& c_end) {
        const int kq = KSPLIT ? ((kpn + 3) / 4 + 7) / 8 * 8 : kpn;
        c_begin = KSPLIT ? min(kpn, wave * kq) : 0;
        c_end = KSPLIT ? min(kpn, c_begin + kq) : kpn;
    };
    auto wload = [&](const uint16_t* const (&wrow)[RPL], int kp0, int c0, int c_end, uint4 (&w)[RPL][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 8;
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                w[q][u] = c < c_end ? load_nt16(wrow[q] + (long long)(kp0 + c) * 8) : uint4{0, 0, 0, 0};
        }
    };
    auto wdot = [&](const uint4 (&w)[RPL][U], int xoff, int c0, int c_end, float (&acc)[RPL][MM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 8;
            if (c < c_end) {
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(&sx[(m * p.kp_chunks + xoff + c) * 8]);
#pragma unroll
                    for (int q = 0; q < RPL; ++q) acc[q][m] = dot8b(w[q][u], xv, acc[q][m]);
                }
            }
        }
    };

    // work item = RPL consecutive units ("super-unit"); KSPLIT: one per workgroup, else one per wave
    const int n_super = (n_units + RPL - 1) / RPL;
    const int sstep = KSPLIT ? gridDim.x : gridDim.x * 4;
    const int su0 = KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
    // The weight stream does not depend on x: the first batch of 16-byte loads of this wave's first item is issued BEFORE x is
    // staged and normalised, and every later batch one step ahead of its use (two register buffers), across item boundaries too.
    uint4 w0[RPL][U], w1[RPL][U];
    const uint16_t* wrow[RPL];
    int cb = 0, ce = 0;           // this wave's chunk range in the first K segment (the same for every unit)
    if (single_piece) {
        wave_range(min(p.canon_chunks, kch), cb, ce);
#pragma unroll
        for (int q = 0; q < RPL; ++q) wrow[q] = unit_row(su0 * RPL + q);
        wload(wrow, 0, cb + kslot, su0 < n_super ? ce : 0, w0);
        stage_x(0, kch);          // the common case: x staged once for every unit of this workgroup
    }

    for (int su = su0; KSPLIT ? su < n_super : (su - wave) < n_super; su += sstep) {
        const bool su_ok = su < n_super;        // non-KSPLIT: the trailing waves of the last workgroup idle but keep the barriers below
#pragma unroll
        for (int q = 0; q < RPL; ++q) wrow[q] = unit_row(su * RPL + q);
        float acc[RPL][MM];
#pragma unroll
        for (int q = 0; q < RPL; ++q)
#pragma unroll
            for (int m = 0; m < MM; ++m) acc[q][m] = 0.f;
        if (single_piece) {
            // x is resident; K is still walked segment by segment (a segment = what fits LDS at M = 8) with the K split over waves
            // inside each segment, so that the fp32 sum order is the one of the staged-in-pieces M = 8 launch
            for (int seg0 = 0; seg0 < kch; seg0 += p.canon_chunks) {
                int sb_, se_;
                wave_range(min(p.canon_chunks, kch - seg0), sb_, se_);
                const int c_end = su_ok ? se_ : 0;
                if (seg0 != 0) wload(wrow, seg0, sb_ + kslot, c_end, w0);    // (the first segment's first batch is already in flight)
                for (int c0 = sb_ + kslot; c0 < c_end; c0 += 16 * U) {
                    wload(wrow, seg0, c0 + 8 * U, c_end, w1);
                    wdot(w0, seg0, c0, c_end, acc);
                    wload(wrow, seg0, c0 + 16 * U, c_end, w0);
                    wdot(w1, seg0, c0 + 8 * U, c_end, acc);
                }
            }
            // next item's first batch goes out before this item's reduction and epilogue
            const int ns = su + sstep;
            const uint16_t* nrow[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) nrow[q] = unit_row(ns * RPL + q);
            wload(nrow, 0, cb + kslot, ns < n_super ? ce : 0, w0);
        } else {
            for (int kp0 = 0; kp0 < kch; kp0 += p.kp_chunks) {
                const int kpn = min(p.kp_chunks, kch - kp0);
                __syncthreads();                 // everyone is done with the previous K piece
                stage_x(kp0, kpn);
                if (su_ok) {
                    int c_begin, c_end;
                    wave_range(kpn, c_begin, c_end);
                    for (int c0 = c_begin + kslot; c0 < c_end; c0 += 8 * U) {
                        wload(wrow, kp0, c0, c_end, w0);
                        wdot(w0, 0, c0, c_end, acc);
                    }
                }
            }
        }
        // ---- reduce the 8 lanes of each row, then (KSPLIT) the 4 waves through LDS ----
#pragma unroll
        for (int q = 0; q < RPL; ++q)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                acc[q][m] += __shfl_xor(acc[q][m], 1, 64);
                acc[q][m] += __shfl_xor(acc[q][m], 2, 64);
                acc[q][m] += __shfl_xor(acc[q][m], 4, 64);
            }
        float* redw = s_red[KSPLIT ? 0 : wave];   // the item's RPL * NR * MM sums (fp32), index (q * NR + j) * MM + m
        if (KSPLIT) {
            if (kslot == 0) {
#pragma unroll
                for (int q = 0; q < RPL; ++q)
#pragma unroll
                    for (int m = 0; m < MM; ++m) s_red[wave][(q * NR + jrow) * MM + m] = acc[q][m];
            }
            __syncthreads();
            if (tid < RPL * NR * MM) {
                const float t = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
                s_red[0][tid] = t;      // each thread reads and writes only its own slot of row 0
            }
            __syncthreads();
        } else {
            if (kslot == 0) {
#pragma unroll
                for (int q = 0; q < RPL; ++q)
#pragma unroll
                    for (int m = 0; m < MM; ++m) redw[(q * NR + jrow) * MM + m] = acc[q][m];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
        const int unit = su * RPL + q;
        const bool unit_ok = su_ok && unit < n_units;
        const float* red = redw + q * NR * MM;
        if (unit_ok && (!KSPLIT || wave == 0)) {
            // ---- epilogue: lane -> (row slot j, sequence m) ----
            if (MODE == GB_PLAIN) {
                if (lane < NR * MM) {
                    const int j = lane / MM, m = lane - j * MM;
                    const int f = unit * 8 + j;
                    if (f < p.N && m < p.M) {
                        float v = red[j * MM + m];
                        if (p.bias) v += bf16_to_f32(p.bias[f]);
                        v = gb_round(v);
                        if (p.res) v += bf16_to_f32(p.res[(long long)m * p.ldr + f]);
                        p.C[(long long)m * p.ldc + f] = f32_to_bf16(v);
                    }
                }
            } else if (MODE == GB_SWIGLU) {
                if (lane < 4 * MM) {
                    const int j = lane / MM, m = lane - j * MM;
                    const int f = unit * 4 + j;
                    if (m < p.M) {
                        const int grow = (f >> 4) * 32 + (f & 15);
                        float g = red[j * MM + m], u = red[(4 + j) * MM + m];
                        if (p.bias) { g += bf16_to_f32(p.bias[grow]); u += bf16_to_f32(p.bias[grow + 16]); }
                        g = gb_round(g);
                        u = gb_round(u);
                        p.C[(long long)m * p.ldc + f] = f32_to_bf16(gb_round(fo1_silu(g)) * u);
                    }
                }
            } else {   // GB_QKV
                if (unit < n_rope) {
                    if (lane < 4 * MM) {
                        const int j = lane / MM, m = lane - j * MM;
                        if (m < p.M) {
                            const int head = unit >> 4, d = (unit & 15) * 4 + j;            // d < 64; partner d + 64
                            const int ra = head * 128 + d, rb_ = ra + 64;
                            float a = red[j * MM + m], b = red[(4 + j) * MM + m];
                            if (p.bias) { a += bf16_to_f32(p.bias[ra]); b += bf16_to_f32(p.bias[rb_]); }
                            a = gb_round(a);                                                 // the bf16 q/k the unfused path stores
                            b = gb_round(b);
                            const int* st = p.state + m * 8;
                            const long long trow = st[1];
                            const float ca = bf16_to_f32(p.cos_t[trow * 128 + d]), sa = bf16_to_f32(p.sin_t[trow * 128 + d]);
                            const float cb2 = bf16_to_f32(p.cos_t[trow * 128 + d + 64]), sb = bf16_to_f32(p.sin_t[trow * 128 + d + 64]);
                            const uint16_t oa = f32_to_bf16(gb_round(a * ca) + gb_round(-b * sa));   // rotate_half, three bf16 roundings
                            const uint16_t ob = f32_to_bf16(gb_round(b * cb2) + gb_round(a * sb));
                            if (head < p.n_q) {
                                p.C[(long long)m * p.ldc + ra] = oa;
                                p.C[(long long)m * p.ldc + rb_] = ob;
                            } else {
                                uint16_t* kc = p.kcache + (long long)(head - p.n_q) * p.kc_head_stride + (long long)st[0] * 128;
                                kc[d] = oa;
                                kc[d + 64] = ob;
                            }
                        }
                    }
                } else {
                    if (lane < NR * MM) {
                        const int j = lane / MM, m = lane - j * MM;
                        if (m < p.M) {
                            const int vrow = (unit - n_rope) * 8 + j;                        // kv_head * 128 + d
                            float v = red[j * MM + m];
                            if (p.bias) v += bf16_to_f32(p.bias[(p.n_q + p.n_kv) * 128 + vrow]);
                            const int* st = p.state + m * 8;
                            p.vtcache[(long long)vrow * p.vt_row_stride + st[0]] = f32_to_bf16(v);
                        }
                    }
                }
            }
        }
        }
        if (KSPLIT) __syncthreads();             // s_red is reused by the next item
        else { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
}

template <int MM, int MODE, bool KS, int RPL>
static int launch_gemv_b(const GemvBParams& p, const char* name, int n_units, hipStream_t st) {
    const size_t smem = (size_t)MM * p.kp_chunks * 16;
    static bool attr = false;
    if (!attr) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_batch_kernel<MM, MODE, KS, RPL>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    // persistent workgroups (grid-stride over work items): x is staged / normalised once per workgroup; <= 4 workgroups per CU
    const int n_super = cdiv(n_units, RPL);
    int grid = KS ? n_super : cdiv(n_super, 4);
    if (grid > 512) {   // about two workgroups per CU, every wave the same number of items
        const int per = cdiv(grid, 512);
        grid = cdiv(grid, per);
    }
    FO1_LAUNCH(name, (double)p.N * p.K * 2.0, (gemv_batch_kernel<MM, MODE, KS, RPL>), dim3(grid), dim3(256), smem, st, p);
    return FO1_OK;
}

extern int g_gemv_profile_shapes;
static int g_gemv_rpl = 0;   // rows per lane: 0 = by M (fo1_gemv_batch_set_rows_per_lane)

template <int MM>
static int dispatch_gemv_b(GemvBParams& p, int mode, hipStream_t st) {
    // K piece staged in LDS: whole K when MM * K * 2 <= 128 KiB, else the smallest number of equal pieces (multiples of 256 chunks)
    const int kch = p.K >> 3;
    int pieces = 1;
    while ((size_t)MM * cdiv(cdiv(kch, pieces), 256) * 256 * 16 > 128 * 1024) ++pieces;
    p.kp_chunks = cdiv(cdiv(kch, pieces), 256) * 256;
    if (p.kp_chunks > kch) p.kp_chunks = kch;
    {   // canonical segment = the piece an 8-sequence launch stages (independent of this launch's M)
        int pc = 1;
        while ((size_t)8 * cdiv(cdiv(kch, pc), 256) * 256 * 16 > 128 * 1024) ++pc;
        p.canon_chunks = cdiv(cdiv(kch, pc), 256) * 256;
        if (p.canon_chunks > kch) p.canon_chunks = kch;
        if (p.kp_chunks < kch) p.kp_chunks = p.canon_chunks;      // staged in pieces: the pieces ARE the canonical segments
    }
    if (p.norm_w && p.kp_chunks < kch) return set_err(FO1_ERR_ARG, "gemv_batch: fused RMSNorm needs K to fit one LDS piece (K=%d, M=%d)", p.K, MM);
    int n_units;
    if (mode == GB_SWIGLU) n_units = p.N / 8;
    else if (mode == GB_QKV) n_units = (p.n_q + p.n_kv) * 16 + p.n_kv * 16;
    else n_units = cdiv(p.N, 8);
    // every CU should stream: one unit per workgroup (K split over its 4 waves) unless that would make more than ~2048 workgroups.
    // (Decided by the shape alone: the K split fixes the fp32 sum order, which must not depend on M.)
    const bool ks = n_units <= 1024;
    // rows per lane: many-unit matrices (gate/up, lm_head) 4 at M = 8, 2 at M = 4 (gate/up 35.6 -> 32.8 us, lm_head 139 -> 120 us
    // at M = 8); the K-split projections stay at 1 — with 2 they have half the workgroups and ran slower (down 22.7 -> 26.5 us)
    constexpr int RA = MM >= 8 ? 4 : (MM >= 4 ? 2 : 1), RB = 1;
    const bool one = g_gemv_rpl == 1;
    char pname[48];
    const char* name = mode == GB_SWIGLU ? "gemv_batch_swiglu" : (mode == GB_QKV ? "gemv_batch_qkv" : "gemv_batch");
    if (profile_enabled() && g_gemv_profile_shapes) {
        snprintf(pname, sizeof pname, "gemv_b m%d %dx%d mode%d ks%d r%d", p.M, p.N, p.K, mode, (int)ks, one ? 1 : (ks ? RB : RA));
        name = pname;
    }
    if (one) {
        if (mode == GB_SWIGLU) return ks ? launch_gemv_b<MM, GB_SWIGLU, true, 1>(p, name, n_units, st) : launch_gemv_b<MM, GB_SWIGLU, false, 1>(p, name, n_units, st);
        if (mode == GB_QKV) return launch_gemv_b<MM, GB_QKV, true, 1>(p, name, n_units, st);
        return ks ? launch_gemv_b<MM, GB_PLAIN, true, 1>(p, name, n_units, st) : launch_gemv_b<MM, GB_PLAIN, false, 1>(p, name, n_units, st);
    }
    if (mode == GB_SWIGLU) return ks ? launch_gemv_b<MM, GB_SWIGLU, true, RB>(p, name, n_units, st) : launch_gemv_b<MM, GB_SWIGLU, false, RA>(p, name, n_units, st);
    if (mode == GB_QKV) return launch_gemv_b<MM, GB_QKV, true, RB>(p, name, n_units, st);
    return ks ? launch_gemv_b<MM, GB_PLAIN, true, RB>(p, name, n_units, st) : launch_gemv_b<MM, GB_PLAIN, false, RA>(p, name, n_units, st);
}

#endif   // FO1_ENABLE_AB (gemv_batch_kernel)

FO1_AB_VAR g_gemv_impl = 1;  // 1 = MFMA skinny GEMM (decode_mfma.hip, M <= 32), 0 = the v_dot2 kernel above (M <= 8)

static int gemv_b_any(GemvBParams& p, int mode, hipStream_t st) {
    // the MFMA kernel moves epilogue operands four features at a time
    const bool quads = p.N % 4 == 0 && (!p.res || (p.ldr % 4 == 0 && ((uintptr_t)p.res & 7) == 0)) && ((uintptr_t)p.bias & 7) == 0 &&
                       ((uintptr_t)p.cos_t & 7) == 0 && ((uintptr_t)p.sin_t & 7) == 0 && ((uintptr_t)p.kcache & 7) == 0;
    if (g_gemv_impl == 1 && quads) return gemv_mfma_any(p, mode, st);
#ifdef FO1_ENABLE_AB
    if (p.M > 8) return set_err(FO1_ERR_ARG, "gemv_batch: the v_dot2 kernel handles M <= 8 (M=%d)", p.M);
    if (p.M == 1) return dispatch_gemv_b<1>(p, mode, st);
    if (p.M == 2) return dispatch_gemv_b<2>(p, mode, st);
    if (p.M <= 4) return dispatch_gemv_b<4>(p, mode, st);
    return dispatch_gemv_b<8>(p, mode, st);
#else
    return set_err(FO1_ERR_ARG, "gemv_batch: operands must be 8-byte aligned with N %% 4 == 0 (N=%d)", p.N);
#endif
}

// ---- argmax over B rows + on-device accept ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_rows_partial_kernel(const uint16_t* __restrict__ x, long long ldx, int n, float* __restrict__ pv,
                                                                  int* __restrict__ pi) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const uint16_t* row = x + (long long)blockIdx.y * ldx;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float v = bf16_to_f32(row[i]);
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
        pv[blockIdx.y * gridDim.x + blockIdx.x] = best;
        pi[blockIdx.y * gridDim.x + blockIdx.x] = bi;
    }
}

// Greedy-search bookkeeping for sequence b given its next token: record it, test the stop rule (EOS / keyword ids, or the
// max_new_tokens budget: HF stops AFTER appending the stop token), advance the device state, publish the token as the next
// step's embedding-gather plan entry.  A finished sequence keeps its state frozen: later steps recompute harmlessly in place.
__device__ __forceinline__ void accept_token(int tok, int* st, int* plan, int* ids_out, int ids_ld, const int* stop_ids, int n_stop, int* done) {
    plan[0] = 0;
    // a finished (or never-started) slot keeps stepping with whatever its rows hold: publish a valid embedding row for it — its
    // logits may be stale or non-finite (the pool skips its attention), and an argmax over NaN rows leaves the index at INT_MAX
    plan[1] = st[3] ? 0 : tok;
    if (st[3]) return;
    const int n = st[4];
    ids_out[n] = tok;
    st[4] = n + 1;
    bool stop = (n + 1 >= st[5]) || (n + 1 >= ids_ld);
    if (n_stop < 0) {       // per-sequence stop sets (the decode pool admits submissions with different sets): row st[6] of a table [sets][1 + 16] = {n, ids...}
        stop_ids += st[6] * 17;
        n_stop = min(stop_ids[0], 16);
        ++stop_ids;
    }
    for (int i = 0; i < n_stop; ++i) stop = stop || (tok == stop_ids[i]);
    if (stop) {
        st[3] = 1;
        atomicAdd(done, 1);
    }
}

// stage 2 of the argmax (one workgroup per row) + accept.  first != 0: the tokens come from `tok_in` (the prefill's argmax)
// instead of the partials, and the state is NOT advanced past the prompt (the first generated token sits at row pos).
__global__ __launch_bounds__(128) void argmax_rows_accept_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int np,
                                                                 const int* __restrict__ tok_in, int* __restrict__ state, int* __restrict__ plan,
                                                                 int* __restrict__ ids_out, int ids_ld, const int* __restrict__ stop_ids, int n_stop,
                                                                 int* __restrict__ done) {
    const int b = blockIdx.x;
    __shared__ float s_v[2];
    __shared__ int s_i[2];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (!tok_in) {
        if ((int)threadIdx.x < np) { best = pv[b * np + threadIdx.x]; bi = pi[b * np + threadIdx.x]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int tok;
        int* st = state + b * 8;
        if (tok_in) {
            tok = tok_in[b];
        } else {
            if (s_v[1] > best || (s_v[1] == best && s_i[1] < bi)) { best = s_v[1]; bi = s_i[1]; }
            tok = bi;
            // the step that produced this token consumed row `pos`: the NEXT fed token goes one row further
            if (!st[3]) { st[0] += 1; st[1] += 1; }
        }
        accept_token(tok, st, plan + 2 * b, ids_out + (long long)b * ids_ld, ids_ld, stop_ids, n_stop, done);
    }
}

// ---- KV relocation: packed prefill rows -> per-sequence decode slots ---------------------------------------------------
// K:  dst[l][h][dst0 + t][:] = src[l][h][src0 + t][:]        (rows of 128 bf16, 16-B pieces)
// VT: dst[l][c][dst0 + t]    = src[l][c][src0 + t]           (c = kv_head*128 + d; src0, dst0 multiples of 4 -> 8-B pieces)
struct RelocSeq { int src0, dst0, len, pad; };
__global__ __launch_bounds__(256) void kv_relocate_kernel(const uint16_t* __restrict__ ksrc, uint16_t* __restrict__ kdst, long long ks_layer,
                                                          long long ks_head, long long kd_layer, long long kd_head,
                                                          const uint16_t* __restrict__ vsrc, uint16_t* __restrict__ vdst, long long vs_layer,
                                                          long long vs_row, long long vd_layer, long long vd_row, const RelocSeq* __restrict__ seqs,
                                                          int n_kv, int n_layers) {
    const RelocSeq s = seqs[blockIdx.y];
    const int layer = blockIdx.z;
    // K: n_kv * len rows of 16 pieces
    const long long nk = (long long)n_kv * s.len * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nk; i += (long long)gridDim.x * blockDim.x) {
        const int pc = (int)(i & 15);
        const long long r = i >> 4;
        const int t = (int)(r % s.len), h = (int)(r / s.len);
        *reinterpret_cast<uint4*>(kdst + layer * kd_layer + h * kd_head + (long long)(s.dst0 + t) * 128 + pc * 8) =
            *reinterpret_cast<const uint4*>(ksrc + layer * ks_layer + h * ks_head + (long long)(s.src0 + t) * 128 + pc * 8);
    }
    // V^T: n_kv * 128 rows, ceil(len / 4) pieces of 4 columns (the tail piece may copy up to 3 columns of the source's padding)
    const int pcs = (s.len + 3) / 4;
    const long long nv = (long long)n_kv * 128 * pcs;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const int pc = (int)(i % pcs);
        const long long c = i / pcs;
        *reinterpret_cast<uint2*>(vdst + layer * vd_layer + c * vd_row + s.dst0 + pc * 4) =
            *reinterpret_cast<const uint2*>(vsrc + layer * vs_layer + c * vs_row + s.src0 + pc * 4);
    }
}

}  // namespace fo1

extern "C" {



#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: test / bench build only
// A/B hook: 1 = one unit (8 weight rows) per lane group whatever M is (the first form of this kernel); 0 = rows per lane by M.
int fo1_gemv_batch_set_rows_per_lane(int rpl) {
    if (rpl != 0 && rpl != 1) return fo1::set_err(FO1_ERR_ARG, "gemv_batch_set_rows_per_lane: %d", rpl);
    fo1::g_gemv_rpl = rpl;
    return FO1_OK;
}


// A/B hook: 1 (default) = MFMA skinny GEMM (decode_mfma.hip, M <= 32); 0 = the v_dot2 streaming kernel (M <= 8);
// 3 = MFMA without any 8-row units; 5 = MFMA with the M <= 8 (HALF) units only, i.e. without round 3's R8 units for 9..32 sequences.
int fo1_gemv_batch_set_impl(int impl) {
    if (impl != 0 && impl != 1 && impl != 3 && impl != 5) return fo1::set_err(FO1_ERR_ARG, "gemv_batch_set_impl: %d", impl);
    fo1::g_gemv_impl = impl & 1;
    fo1::g_gemv_half = (impl & 2) ? 0 : ((impl & 4) ? 1 : 3);
    return FO1_OK;
}
#endif   // FO1_ENABLE_AB

// Batched decode projection: C[M<=16, N] = epilogue(rmsnorm?(x) @ W^T), weights streamed once for all M rows.
// mode 0: bias -> bf16 -> + residual;  mode 1: interleaved SwiGLU (C has N/2 columns);  mode 2: fused QKV:
//   bias -> bf16 -> mRoPE (table row state[m][1]) -> rotated q rows to C[m, 0 : n_q*128), rotated K row to kcache[kv][state[m][0]],
//   V to the V^T cache column state[m][0].
int fo1_gemv_batch_bf16(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, int ldr, void* C, int ldc,
                        int M, int N, int K, int mode, const void* norm_weight, float norm_eps, int n_q_heads, int n_kv_heads,
                        const void* cos_table, const void* sin_table, const int32_t* state, void* kcache, long long kcache_head_stride,
                        void* vtcache, long long vt_row_stride, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && W && (C || mode == 2), "gemv_batch: NULL operand");
    FO1_CHECK_ARG(M >= 1 && M <= 32 && N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "gemv_batch: bad shape M=%d (1..32) N=%d K=%d", M, N, K);
    FO1_CHECK_ARG(mode >= 0 && mode <= 2, "gemv_batch: mode %d", mode);
    FO1_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)norm_weight & 15) == 0, "gemv_batch: misaligned operand");
    if (mode == 1) FO1_CHECK_ARG(N % 32 == 0 && residual == nullptr, "gemv_batch: SwiGLU needs N %% 32 == 0 and no residual");
    if (mode == 2) {
        FO1_CHECK_ARG(n_q_heads > 0 && n_kv_heads > 0 && N == (n_q_heads + 2 * n_kv_heads) * 128, "gemv_batch: QKV mode needs N = (n_q + 2 n_kv) * 128");
        FO1_CHECK_ARG(cos_table && sin_table && state && kcache && vtcache && C && residual == nullptr, "gemv_batch: QKV mode operands");
    }
    GemvBParams p;
    p.X = (const uint16_t*)x; p.W = (const uint16_t*)W; p.bias = (const uint16_t*)bias; p.res = (const uint16_t*)residual; p.C = (uint16_t*)C;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
    p.norm_w = (const uint16_t*)norm_weight; p.norm_eps = norm_eps; p.kp_chunks = 0;
    p.n_q = n_q_heads; p.n_kv = n_kv_heads; p.cos_t = (const uint16_t*)cos_table; p.sin_t = (const uint16_t*)sin_table; p.state = (const int*)state;
    p.kcache = (uint16_t*)kcache; p.kc_head_stride = kcache_head_stride; p.vtcache = (uint16_t*)vtcache; p.vt_row_stride = vt_row_stride;
    return gemv_b_any(p, mode, (hipStream_t)stream);
}

// The o-projection of a decode step at <= 2 sequences with the split-KV attention combine in its prologue: C[M, N] = bf16(bf16(x W^T) + residual),
// x[m, head * 128 + d] = sum_s exp(m_s - M) O_s[d] / sum_s exp(m_s - M) l_s over the chunks of fo1_attention_decode_batch_partials_bf16 — the value
// attn_decode_combine_kernel writes, bit for bit (one shared routine), without its launch.  K = n_q_heads * 128 <= 2048, N <= 4096.
int fo1_gemv_attn_combine_bf16(const float* part, long long part_seq_stride, const int32_t* state, int kv_chunk, int n_q_heads, int n_kv_heads,
                               const void* W, int ldw, const void* residual, int ldr, void* C, int ldc, int M, int N, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(part && state && W && C, "gemv_attn_combine: NULL operand");
    const int K = n_q_heads * 128;
    FO1_CHECK_ARG(M >= 1 && M <= 2 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && n_q_heads / n_kv_heads <= 16 && K <= 2048 && N > 0 && N <= 4096 && N % 4 == 0,
                  "gemv_attn_combine: M=%d (1..2) heads %d / %d (K = heads x 128 <= 2048) N=%d (<= 4096)", M, n_q_heads, n_kv_heads, N);
    FO1_CHECK_ARG(kv_chunk >= 64 && kv_chunk % 64 == 0 && ldw % 8 == 0 && ldw >= K && ((uintptr_t)W & 15) == 0 && ((uintptr_t)part & 7) == 0, "gemv_attn_combine: bad layout");
    FO1_CHECK_ARG(ldc >= N && (residual == nullptr || (ldr % 4 == 0 && ldr >= N && ((uintptr_t)residual & 7) == 0)), "gemv_attn_combine: residual / output layout");
    GemvBParams p;
    p.X = (const uint16_t*)W; p.ldx = 0;      // (the staged rows come from the partials; the plain x loads of the kernel's prologue read valid memory and are ignored)
    p.W = (const uint16_t*)W; p.bias = nullptr; p.res = (const uint16_t*)residual; p.C = (uint16_t*)C;
    p.M = M; p.N = N; p.K = K; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
    p.norm_w = nullptr; p.norm_eps = 0.f; p.kp_chunks = 0;
    p.n_q = p.n_kv = 0; p.cos_t = p.sin_t = nullptr; p.state = nullptr; p.kcache = p.vtcache = nullptr; p.kc_head_stride = p.vt_row_stride = 0;
    p.attn_part = part; p.attn_part_seq_stride = part_seq_stride; p.attn_state = (const int*)state;
    p.attn_chunk = kv_chunk; p.attn_n_kv = n_kv_heads; p.attn_group = n_q_heads / n_kv_heads;
    return gemv_mfma_any(p, GB_PLAIN, (hipStream_t)stream);
}

// Greedy pick for B logits rows + the on-device bookkeeping of one decode step (see accept_token).
// logits == NULL: accept `first_tokens` (int32[B], the prefill's argmax) without advancing the positions.
// scratch: 2 * 128 * B * 4 bytes.  plan: int32[B][2] gather plan for the next step's embedding rows.  done: int32 counter of
// finished sequences (never reset here).
int fo1_decode_argmax_accept(const void* logits, long long ld_logits, int n_vocab, int B, const int32_t* first_tokens, int32_t* state,
                             int32_t* plan, int32_t* ids_out, int ids_ld, const int32_t* stop_ids, int n_stop, int32_t* done, void* scratch,
                             void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(state && plan && ids_out && done && B >= 1 && B <= 256 && ids_ld > 0 && n_stop >= -1 && (n_stop == 0 || stop_ids), "decode_accept: bad arguments");
    FO1_CHECK_ARG((logits != nullptr) != (first_tokens != nullptr), "decode_accept: exactly one of logits / first_tokens");
    hipStream_t st = (hipStream_t)stream;
    float* pv = (float*)scratch;
    int* pi = (int*)(pv + 128 * B);
    if (logits) {
        FO1_CHECK_ARG(scratch && n_vocab > 0, "decode_accept: scratch / vocab");
        FO1_LAUNCH("argmax_rows", (double)B * n_vocab * 2.0, argmax_rows_partial_kernel, dim3(128, B), dim3(256), 0, st, (const uint16_t*)logits,
                   ld_logits, n_vocab, pv, pi);
    }
    FO1_LAUNCH("argmax_accept", 1024.0 * B, argmax_rows_accept_kernel, dim3(B), dim3(128), 0, st, (const float*)pv, (const int*)pi, 128,
               (const int*)first_tokens, (int*)state, (int*)plan, (int*)ids_out, ids_ld, (const int*)stop_ids, n_stop, (int*)done);
    return FO1_OK;
}

// Copies every sequence's K rows and V^T columns from the packed prefill positions to its decode slot, all layers, one launch.
// seqs: device int32[B][4] = {src0, dst0, len, 0}; src0 and dst0 multiples of 4.
int fo1_kv_relocate(const void* ksrc, void* kdst, long long ks_layer, long long ks_head, long long kd_layer, long long kd_head,
                    const void* vsrc, void* vdst, long long vs_layer, long long vs_row, long long vd_layer, long long vd_row,
                    const int32_t* seqs, int B, int max_len, int n_kv_heads, int n_layers, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(ksrc && kdst && vsrc && vdst && seqs && B >= 1 && n_kv_heads >= 1 && n_layers >= 1 && max_len >= 1, "kv_relocate: bad arguments");
    const long long work = (long long)n_kv_heads * max_len * 16;
    int gx = (int)((work + 255) / 256);
    if (gx > 64) gx = 64;
    FO1_LAUNCH("kv_relocate", (double)B * n_layers * n_kv_heads * max_len * 128 * 8.0, kv_relocate_kernel, dim3(gx, B, n_layers), dim3(256), 0,
               (hipStream_t)stream, (const uint16_t*)ksrc, (uint16_t*)kdst, ks_layer, ks_head, kd_layer, kd_head, (const uint16_t*)vsrc,
               (uint16_t*)vdst, vs_layer, vs_row, vd_layer, vd_row, (const RelocSeq*)seqs, n_kv_heads, n_layers);
    return FO1_OK;
}

}  // extern "C"
