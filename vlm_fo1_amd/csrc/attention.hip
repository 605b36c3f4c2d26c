// attention.hip — fused softmax(QK^T)V on MFMA for gfx950: the ViT's windowed/full varlen
// attention (head dim 80), the LLM's causal GQA prefill (head dim 128) and DaViT's 12x12 window
// attention (head dim 32).  Replaces flash_attn_varlen_func / _flash_attention_forward /
// F.scaled_dot_product_attention at reference modeling_qwen2_5_vl.py:205,319,895,990 and the
// explicit softmax(q k^T) v at modeling_davit.py:262-270.
//
// One workgroup = 4 waves = one work item (<= 64 queries of one segment) x one query head; each
// wave owns 16 queries.  Per 64-key tile: K rows and V^T rows are staged once in LDS and shared by
// the 4 waves.  Scores are computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_16x16x32_bf16 with a = K,
// b = Q) so each lane owns one query column: the online-softmax statistics are lane-local plus two
// cross-lane-group shuffles, and the exponentiated P values are already in the B-operand layout of
// the second MFMA (O^T = V^T P^T), whose accumulator rows are head-dim indices and whose column is
// the same query - so the per-query rescale needs no data movement either.  P is rounded to bf16
// before the PV product (as flash-attention does); accumulation, max and sum are fp32.
//
// V is consumed as V^T ([kv_head*HD + d][key]) so that a lane's 8 k-slots are two 8-byte LDS reads;
// the transposed copy is written once per layer by fo1_transpose_bf16 (rope.hip).
#include "common.h"
#include "ab.h"
#include "decode_common.h"

namespace fo1 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnItem {
    int q_start, q_end;    // queries [q_start, q_end), q_end - q_start <= q_block (16 per wave)
    int kv_start, kv_end;  // keys [kv_start, kv_end); causal: additionally key <= query (same index space)
};

struct AttnParams {
    const uint16_t* Q; long long q_tok, q_head;      // element strides
    const uint16_t* K; long long k_tok, k_head;
    const uint16_t* VT; long long vt_row;            // V^T row stride (elements); row = kv_head*HD + d
    uint16_t* O; long long o_tok, o_head;
    const AttnItem* items;
    const int2* items2;                              // optional, per item: a SECOND key range [x, y) every query of the item sees in full, walked
                                                     // before the item's own range — the shared prompt prefix of several prompts over one
                                                     // image (its rows precede the item's in the index space, so `key <= query` holds)
    int n_items, Hq, group;                          // group = Hq / Hkv
    float scale;
    int causal;
    const int* q_row_base;                           // optional: Q/O row = query index - *q_row_base (decode graphs)
    // split-KV decode (PARTIAL): each item covers one KV chunk; unnormalised fp32 O and (m, l) go to `part`
    float* part;                                     // [n_items][Hq][16 slots][HD + 2]
    const int* dyn_kv_len;                           // optional: kv_end/kv_start of item i derived on device: chunk i of *dyn_kv_len keys
    int kv_chunk;
    int q_range_end;                                 // PARTIAL: number of query heads per KV head
    int grid_batch;                                  // PARTIAL with seq_state: > 0 = 1-D grid of n_items x Hq x grid_batch workgroups walked sequence-fastest (see the kernel)
    int part_tiles;                                  // PARTIAL, > 1: an item walks part_tiles 64-key tiles (kv_chunk = 64 x part_tiles) and writes ONE PARTIAL PER TILE, each
                                                     // exactly what a one-tile item writes (state reset between tiles): the workgroup count and the load / compute overlap of a
                                                     // long item with the partial sums of 64-key splits — a sequence's attention does not depend on the choice (round 6)
    // batched decode (PARTIAL): blockIdx.z = sequence; keys [state[z][2], state[z][0]] (decode.hip state layout)
    const int* seq_state;
    long long q_seq_stride;                          // Q elements between sequences
    long long part_seq_stride;                       // floats between sequences in `part`
    // window attention with an additive bias (Swin: relative position bias + shifted-window mask, backbone/swin.py:150-175):
    // bias fp32 [Hq][wlen][wlen], indexed by the query / key position inside the item's kv range (= one window of wlen tokens)
    const float* bias;
    int wlen;
    int sw_ws, sw_shift, sw_nwy, sw_nwx;             // sw_shift > 0: -100 between tokens of different shift regions (:446-466)
};

template <int HD, int NW, bool PARTIAL = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(const AttnParams p) {
    constexpr int NT = NW * 64;
    constexpr int HDP = (HD + 31) / 32 * 32;  // head dim padded to the MFMA K step
    constexpr int NC = HDP / 32;              // 32-wide d chunks for QK^T
    constexpr int NDB = HD / 16;              // 16-wide d blocks for PV
    constexpr int KB = 64;                    // keys per tile
    constexpr int LDKR = HDP + 8;             // K tile row pitch (elements)
    constexpr int LDVT = KB + 4;              // V^T tile row pitch (elements): 136 B rows, conflict-free b64 reads
    __shared__ __attribute__((aligned(16))) uint16_t sK[KB * LDKR];
    __shared__ __attribute__((aligned(16))) uint16_t sVT[HD * LDVT];

    // block coordinates.  Batched decode (PARTIAL with seq_state) launches a 1-D grid walked SEQUENCE-FASTEST: (chunk, kv head, sequence) with the
    // chunk slowest — the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, and with the chunk fastest the few chunks of a
    // context that exist (11 of 32 at 700 keys in a 2048-key bucket) always landed on the same XCDs whenever the chunk count divided by 8
    // (round 6: 16.1 -> 10.1 us per layer at 25 sequences); the empty chunks now come last.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (PARTIAL && p.seq_state && p.grid_batch > 0) {
        const int lin = blockIdx.x;
        bz = lin % p.grid_batch;
        by = (lin / p.grid_batch) % p.Hq;
        bx = lin / (p.grid_batch * p.Hq);
    }
    AttnItem it;
    if (!PARTIAL) it = p.items[bx];
    const uint16_t* Qb = p.Q;
    float* partb = p.part;
    if (PARTIAL) {
        it.q_start = 0;
        it.q_end = p.q_range_end;   // chunk blockIdx.x of the sequence's keys; empty chunks leave (m, l) = (-inf, 0)
        int kv0 = 0, kv_len;
        if (p.seq_state) {
            const int* st = p.seq_state + bz * 8;
            if (st[3]) {
                // finished (or empty pool slot): nobody reads this sequence's tokens, but its attention row still flows through the
                // o-projection into the slot's frozen K / V^T row of every later layer — it must be FINITE (0 x NaN would poison a later
                // occupant whose keys come within the 4-key V^T piece of that row).  One-chunk form: the row is written
                // here, as zeros; split form: attn_decode_combine_kernel writes it.
                if (p.O != nullptr && bx == 0 && threadIdx.x < 64) {
                    const int ql = threadIdx.x & 15, g = threadIdx.x >> 4;
                    if (ql < p.q_range_end) {
                        uint16_t* op = p.O + (long long)bz * p.o_tok + ((long long)by * p.q_range_end + ql) * p.o_head;
#pragma unroll
                        for (int db = 0; db < HD / 16; ++db) *reinterpret_cast<uint2*>(op + db * 16 + g * 4) = uint2{0u, 0u};
                    }
                }
                return;
            }
            kv0 = st[2];
            kv_len = st[0] + 1;
            Qb += (long long)bz * p.q_seq_stride;
            partb += (long long)bz * p.part_seq_stride;
        } else {
            kv_len = *p.dyn_kv_len;
        }
        it.kv_start = kv0 + bx * p.kv_chunk;
        it.kv_end = min(kv_len, it.kv_start + p.kv_chunk);
        if (it.kv_start >= kv_len) return;
    }
    const int h = by, kvh = h / p.group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, g = lane >> 4;
    const int q_idx = it.q_start + wave * 16 + ql;          // this lane's query (score column)
    const bool q_ok = q_idx < it.q_end;
    const int q_base = p.q_row_base ? *p.q_row_base : 0;
    const int q_ld = (q_ok ? q_idx : it.q_end - 1) - q_base;
    // additive bias (and shift regions) of this window: row of the lane's query, region id of every token of the window
    __shared__ unsigned char s_reg[256];
    const float* bias_row = nullptr;
    int rid_q = 0;
    if (!PARTIAL && p.bias) {
        const int ql_loc = (q_ok ? q_idx : it.q_end - 1) - it.kv_start;
        bias_row = p.bias + ((long long)h * p.wlen + ql_loc) * p.wlen;
        if (p.sw_shift > 0) {
            const int win = (it.kv_start / p.wlen) % (p.sw_nwy * p.sw_nwx);
            const int wy = win / p.sw_nwx, wx = win - wy * p.sw_nwx;
            const int Hp = p.sw_nwy * p.sw_ws, Wp = p.sw_nwx * p.sw_ws;
            for (int t = tid; t < p.wlen; t += NT) {
                const int y = wy * p.sw_ws + t / p.sw_ws, x = wx * p.sw_ws + t % p.sw_ws;
                const int ry = y < Hp - p.sw_ws ? 0 : (y < Hp - p.sw_shift ? 1 : 2);
                const int rx = x < Wp - p.sw_ws ? 0 : (x < Wp - p.sw_shift ? 1 : 2);
                s_reg[t] = (unsigned char)(ry * 3 + rx);
            }
            __syncthreads();
            rid_q = s_reg[ql_loc];
        }
    }

    // Q fragments (B operand): lane (query ql, k-group g) holds d = c*32 + g*8 .. +8
    bf16x8 qf[NC];
    {
        const uint16_t* qp = Qb + (long long)q_ld * p.q_tok + (long long)h * p.q_head;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int d0 = c * 32 + g * 8;
            uint4 v = uint4{0, 0, 0, 0};
            if (d0 < HD) v = *reinterpret_cast<const uint4*>(qp + d0);
            qf[c] = *reinterpret_cast<bf16x8*>(&v);
        }
    }

    f32x4 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;       // m_run in base-2 units: max of c1 * score
    const float sl2 = p.scale * 1.44269504088896f;

    int kv_hi = it.kv_end;
    if (p.causal && it.q_end < kv_hi) kv_hi = it.q_end;  // keys beyond the last query are never attended
    const uint16_t* Kb = p.K + (long long)kvh * p.k_head;
    const uint16_t* VTb = p.VT + (long long)kvh * HD * p.vt_row;

    // K / V^T tiles are prefetched into registers one tile ahead (issue-early / write-late): the global-load
    // latency of tile t+1 hides under the MFMAs + softmax of tile t.
    constexpr int KCH = HDP / 8;                       // 16-B chunks per K row
    constexpr int NKR = (KB * KCH + NT - 1) / NT;      // K chunks per thread
    constexpr int NVR = (HD * (KB / 4) + NT - 1) / NT; // V^T 8-byte pieces per thread
    uint4 rk[NKR];
    uint2 rv[NVR];
    auto gload = [&](int k0, int kv_hi) {
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = tid + i * NT;
            const int r = q / KCH, c = q - r * KCH;
            rk[i] = uint4{0, 0, 0, 0};
            if (q < KB * KCH && k0 + r < kv_hi && c * 8 < HD)
                rk[i] = *reinterpret_cast<const uint4*>(Kb + (long long)(k0 + r) * p.k_tok + c * 8);
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = tid + i * NT;
            const int d = q / (KB / 4), c = q - d * (KB / 4);
            rv[i] = uint2{0, 0};
            if (q < HD * (KB / 4) && k0 + c * 4 < kv_hi)
                rv[i] = *reinterpret_cast<const uint2*>(VTb + (long long)d * p.vt_row + k0 + c * 4);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = tid + i * NT;
            const int r = q / KCH, c = q - r * KCH;
            if (q < KB * KCH) *reinterpret_cast<uint4*>(&sK[r * LDKR + c * 8]) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = tid + i * NT;
            const int d = q / (KB / 4), c = q - d * (KB / 4);
            if (q < HD * (KB / 4)) *reinterpret_cast<uint2*>(&sVT[d * LDVT + c * 4]) = rv[i];
        }
    };
    // tile walk: the optional second range first (the shared prefix), then the item's own range
    const int own_hi = kv_hi;
    int2 r2 = int2{0, 0};
    if (!PARTIAL && p.items2) r2 = p.items2[bx];
    const bool has2 = r2.y > r2.x;
    int k0 = has2 ? r2.x : it.kv_start;
    kv_hi = has2 ? r2.y : own_hi;
    bool in2 = has2;
    if (k0 < kv_hi) gload(k0, kv_hi);
    else if (in2) { in2 = false; k0 = it.kv_start; kv_hi = own_hi; if (k0 < kv_hi) gload(k0, kv_hi); }
    for (; k0 < kv_hi;) {
        __syncthreads();  // previous tile fully consumed
        swrite();
        __syncthreads();
        // the tile after this one (possibly the first tile of the item's own range)
        int nk0 = k0 + KB, nhi = kv_hi;
        bool nin2 = in2;
        if (nk0 >= kv_hi && in2) { nk0 = it.kv_start; nhi = own_hi; nin2 = false; }
        if (nk0 < nhi) gload(nk0, nhi);

        // ---- S^T = K Q^T : 4 key sub-tiles of 16 ----
        // Every fragment read of the tile is issued before the first MFMA (and the V^T fragments before the softmax): left to itself
        // hipcc ping-pongs two fragment registers, "ds_read, s_waitcnt lgkmcnt(1), mfma" 4 x NC times, and the loop runs at LDS
        // latency (MFMA busy 8 %, profiles/r02_mfma_busy.json).  sched_barrier(0) pins the read block in front of the MFMA block.
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {         // two batches of 2 x NC fragments: the whole tile at once costs an occupancy step
            bf16x8 kf[2][NC];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[kt][c] = *reinterpret_cast<const bf16x8*>(&sK[((kh * 2 + kt) * 16 + ql) * LDKR + c * 32 + g * 8]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    s[kh * 2 + kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[c], s[kh * 2 + kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // V^T fragments of the first key half: in flight while the softmax runs.  a-operand k-slots: j<4 -> key (2*half)*16 + g*4 + j ;
        // j>=4 -> key (2*half+1)*16 + g*4 + (j-4)
        uint2 vlo[NDB], vhi[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + g * 4];
            vlo[db] = *reinterpret_cast<const uint2*>(vr);
            vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        // s[kt][r] = score(key = k0 + kt*16 + g*4 + r, query = q_idx), raw (unscaled).
        // The softmax runs in base 2 on c1 * s (c1 = scale * log2 e): one v_fma + one v_exp per score, the mask only on the tiles that
        // need one (the last tile of the range, the causal diagonal; wave-uniform), no bf16 round trip for the row sum (flash-attention
        // sums the fp32 probabilities too) — the VALU, not the MFMA, bounded this loop (profiles/README.md, MFMA busy 8 %).
        float c1 = sl2;
        const bool need_mask = (k0 + KB > kv_hi) || (p.causal && k0 + KB - 1 > it.q_start + wave * 16);
        if (!PARTIAL && bias_row) {
            c1 = 1.44269504088896f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool ok = key < kv_hi && (!p.causal || key <= q_idx);
                    const int kl = ok ? key - it.kv_start : 0;
                    float v = s[kt][r] * p.scale + bias_row[kl];
                    if (p.sw_shift > 0 && s_reg[kl] != rid_q) v += -100.0f;
                    s[kt][r] = ok ? v : -INFINITY;
                }
        } else if (need_mask) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool ok = key < kv_hi && (!p.causal || key <= q_idx);
                    s[kt][r] = ok ? s[kt][r] : -INFINITY;
                }
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3])), fmaxf(fmaxf(s[3][0], s[3][1]), fmaxf(s[3][2], s[3][3]))));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c1);               // running max of c1 * s (c1 > 0)
        // a query with no valid key yet (only possible for padding lanes) keeps everything at zero
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0 (o and l are zero then anyway)
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t w[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = half * 2 + t;
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c1, -m_use));   // masked: exp2(-inf) = 0
                    psum += e[r];
                }
                w[t * 2 + 0] = pack_bf16x2(e[0], e[1]);
                w[t * 2 + 1] = pack_bf16x2(e[2], e[3]);
            }
            uint4 pk = uint4{w[0], w[1], w[2], w[3]};
            pf[half] = *reinterpret_cast<bf16x8*>(&pk);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        const bool rescale = __any(alpha != 1.0f);               // wave-uniform: later tiles rarely move the maximum
        // ---- O^T = alpha * O^T + V^T P^T ----
        if (rescale) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) { o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha; }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 vk[NDB];
#pragma unroll
            for (int db = 0; db < NDB; ++db) vk[db] = uint4{vlo[db].x, vlo[db].y, vhi[db].x, vhi[db].y};
            if (half == 0) {                     // second key half: requested before the first half's MFMAs
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + 32 + g * 4];
                    vlo[db] = *reinterpret_cast<const uint2*>(vr);
                    vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&vk[db]), pf[half], o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PARTIAL && p.part_tiles > 1) {
            // per-tile partial (wave-uniform branch): chunk index = item * part_tiles + tile of the item
            if (wave == 0 && q_ok) {
                const long long ci = (long long)bx * p.part_tiles + (k0 - it.kv_start) / KB;
                float* pr = partb + ((ci * p.Hq + h) * 16 + ql) * (HD + 2);
#pragma unroll
                for (int db = 0; db < NDB; ++db)
                    *reinterpret_cast<float4*>(pr + db * 16 + g * 4) = float4{o[db][0], o[db][1], o[db][2], o[db][3]};
                if (g == 0) { pr[HD] = m_run * 0.6931471805599453f; pr[HD + 1] = l_run; }
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run = -INFINITY;
            l_run = 0.f;
        }
        k0 = nk0;
        kv_hi = nhi;
        in2 = nin2;
    }
    if (PARTIAL && p.part_tiles > 1) return;

    if (PARTIAL) {
        if (p.O != nullptr) {
            // batched decode whose chunk covers the whole context (ONE item per sequence and KV head): the normalised rows go straight to
            // the output — the value attn_decode_combine_kernel computes from a single partial (w = exp(0) = 1: num / den = o / l) —
            // and the combine launch is not made.  o_tok = the output's sequence stride, o_head = HD.
            if (wave == 0 && q_ok) {
                uint16_t* op = p.O + (long long)bz * p.o_tok + ((long long)h * p.q_range_end + ql) * p.o_head;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    uint2 w;
                    w.x = pack_bf16x2(l_run > 0.f ? o[db][0] / l_run : 0.f, l_run > 0.f ? o[db][1] / l_run : 0.f);
                    w.y = pack_bf16x2(l_run > 0.f ? o[db][2] / l_run : 0.f, l_run > 0.f ? o[db][3] / l_run : 0.f);
                    *reinterpret_cast<uint2*>(op + db * 16 + g * 4) = w;
                }
            }
            return;
        }
        if (wave == 0 && q_ok) {
            float* pr = partb + (((long long)bx * p.Hq + h) * 16 + ql) * (HD + 2);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                *reinterpret_cast<float4*>(pr + db * 16 + g * 4) = float4{o[db][0], o[db][1], o[db][2], o[db][3]};
            if (g == 0) { pr[HD] = m_run * 0.6931471805599453f; pr[HD + 1] = l_run; }   // the combine kernel works in natural units
        }
        return;
    }
    // o[db][r] = O[query q_idx][d = db*16 + g*4 + r]
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        uint16_t* op = p.O + (long long)(q_idx - q_base) * p.o_tok + (long long)h * p.o_head;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 w;
            w.x = pack_bf16x2(o[db][0] * inv, o[db][1] * inv);
            w.y = pack_bf16x2(o[db][2] * inv, o[db][3] * inv);
            *reinterpret_cast<uint2*>(op + db * 16 + g * 4) = w;
        }
    }
}

// ---- single-tile items in a software pipeline (round 6): the ViT's windows (modeling_qwen2_5_vl.py:172-209) ------------------------------------
// attn_fwd_kernel spends a 64-token window as one workgroup: load K / V^T / Q -> S -> softmax -> PV -> store, one dependent chain of ~9 us with
// nothing to overlap (a window is ONE key tile: the kernel's next-tile prefetch has nothing to fetch), three workgroups per CU.  Here a workgroup
// walks IPW consecutive items of the list for its head and requests item i + 1's K / V^T tile and Q fragments right after item i's tile is in
// LDS: the loads fly under item i's arithmetic.  Same fragment algebra, same instruction order per item as attn_fwd_kernel<HD, 4> on a
// one-tile item (running maximum from -inf, o = 0 * alpha + V^T P^T): bit-identical outputs (tests/test_ops_gpu.py).  Items: non-causal,
// at most 64 keys, q range inside the kv range; stores through a buffer descriptor (no branch around them: see dwconv3x3_ln_run_kernel).
template <int HD, int IPW>
__global__ __launch_bounds__(256, 2) void attn_win1_kernel(const AttnParams p, uint32_t o_bytes) {
    constexpr int NT = 256;
    constexpr int HDP = (HD + 31) / 32 * 32, NC = HDP / 32, NDB = HD / 16, KB = 64;
    constexpr int LDKR = HDP + 8, LDVT = KB + 4;
    __shared__ __attribute__((aligned(16))) uint16_t sK[KB * LDKR];
    __shared__ __attribute__((aligned(16))) uint16_t sVT[HD * LDVT];
    const int h = blockIdx.y, kvh = h / p.group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, g = lane >> 4;
    const int item0 = blockIdx.x * IPW;
    if (item0 >= p.n_items) return;
    const uint16_t* Kb = p.K + (long long)kvh * p.k_head;
    const uint16_t* VTb = p.VT + (long long)kvh * HD * p.vt_row;
    const uint16_t* Qh = p.Q + (long long)h * p.q_head;
    typedef __attribute__((ext_vector_type(2))) unsigned int wv2u;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.O, 0, o_bytes, 0x00020000);        // <= 2 GiB (host-checked)
    constexpr int KCH = HDP / 8;
    constexpr int NKR = (KB * KCH + NT - 1) / NT;
    constexpr int NVR = (HD * (KB / 4) + NT - 1) / NT;
    uint4 rk[NKR];
    uint2 rv[NVR];
    uint4 qn[NC];
    // every load unconditional from a clamped address (rows / keys past the item: its last row / last 4-key piece — finite, masked below)
    auto gload = [&](const AttnItem& it) {
        const int k0 = it.kv_start, nk = it.kv_end - it.kv_start;
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = min(tid + i * NT, KB * KCH - 1);
            const int r = q / KCH, c = q - r * KCH;
            rk[i] = *reinterpret_cast<const uint4*>(Kb + (long long)(k0 + min(r, nk - 1)) * p.k_tok + min(c * 8, HD - 8));
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = min(tid + i * NT, HD * (KB / 4) - 1);
            const int d = q / (KB / 4), c = q - d * (KB / 4);
            rv[i] = *reinterpret_cast<const uint2*>(VTb + (long long)d * p.vt_row + k0 + min(c * 4, max((nk - 1) & ~3, 0)));
        }
        const int qi = min(it.q_start + wave * 16 + ql, it.q_end - 1);
        const uint16_t* qp = Qh + (long long)qi * p.q_tok;
#pragma unroll
        for (int c = 0; c < NC; ++c) qn[c] = *reinterpret_cast<const uint4*>(qp + min(c * 32 + g * 8, HD - 8));
    };
    AttnItem it = p.items[item0];
    gload(it);
    const float sl2 = p.scale * 1.44269504088896f;
#pragma unroll
    for (int rep = 0; rep < IPW; ++rep) {
        if (item0 + rep >= p.n_items) break;                   // workgroup-uniform
        const int nk = it.kv_end - it.kv_start;
        // this item's registers -> LDS / fragments (the zero padding attn_fwd_kernel's guarded loads produce: rows past the item, columns past HD)
        bf16x8 qf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint4 v = qn[c];
            if (c * 32 + g * 8 >= HD) v = uint4{0, 0, 0, 0};
            qf[c] = *reinterpret_cast<bf16x8*>(&v);
        }
        __syncthreads();                                       // the previous item's tile is consumed
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = tid + i * NT;
            const int r = q / KCH, c = q - r * KCH;
            const bool ok = q < KB * KCH && r < nk && c * 8 < HD;
            if (q < KB * KCH) *reinterpret_cast<uint4*>(&sK[r * LDKR + c * 8]) = ok ? rk[i] : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = tid + i * NT;
            const int d = q / (KB / 4), c = q - d * (KB / 4);
            const bool ok = q < HD * (KB / 4) && c * 4 < nk;
            if (q < HD * (KB / 4)) *reinterpret_cast<uint2*>(&sVT[d * LDVT + c * 4]) = ok ? rv[i] : uint2{0, 0};
        }
        __syncthreads();
        AttnItem nx = it;
        if (rep + 1 < IPW && item0 + rep + 1 < p.n_items) {    // the next item's tile and queries: in flight under this item's arithmetic
            nx = p.items[item0 + rep + 1];
            gload(nx);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T = K Q^T (attn_fwd_kernel's block structure) ----
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8 kf[2][NC];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[kt][c] = *reinterpret_cast<const bf16x8*>(&sK[((kh * 2 + kt) * 16 + ql) * LDKR + c * 32 + g * 8]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    s[kh * 2 + kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[c], s[kh * 2 + kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint2 vlo[NDB], vhi[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + g * 4];
            vlo[db] = *reinterpret_cast<const uint2*>(vr);
            vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nk < KB) {                                         // workgroup-uniform
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = kt * 16 + g * 4 + r < nk ? s[kt][r] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3])), fmaxf(fmaxf(s[3][0], s[3][1]), fmaxf(s[3][2], s[3][3]))));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = mx * sl2;                          // fmaxf(-inf, mx * c1)
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t w[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = half * 2 + t;
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sl2, -m_use));
                    psum += e[r];
                }
                w[t * 2 + 0] = pack_bf16x2(e[0], e[1]);
                w[t * 2 + 1] = pack_bf16x2(e[2], e[3]);
            }
            uint4 pk = uint4{w[0], w[1], w[2], w[3]};
            pf[half] = *reinterpret_cast<bf16x8*>(&pk);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        const float l_run = psum;                              // 0 * alpha + psum
        f32x4 o[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 vk[NDB];
#pragma unroll
            for (int db = 0; db < NDB; ++db) vk[db] = uint4{vlo[db].x, vlo[db].y, vhi[db].x, vhi[db].y};
            if (half == 0) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + 32 + g * 4];
                    vlo[db] = *reinterpret_cast<const uint2*>(vr);
                    vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&vk[db]), pf[half], o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // o[db][r] = O[query][d = db * 16 + g * 4 + r]
        const int q_idx = it.q_start + wave * 16 + ql;
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        const uint32_t obase = q_idx < it.q_end ? (uint32_t)(((long long)q_idx * p.o_tok + (long long)h * p.o_head + g * 4) * 2) : 0xC0000000u;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            __builtin_amdgcn_raw_buffer_store_b64(wv2u{pack_bf16x2(o[db][0] * inv, o[db][1] * inv), pack_bf16x2(o[db][2] * inv, o[db][3] * inv)}, rs_o,
                                                  obase + (uint32_t)(db * 32), 0, 0);
        it = nx;
    }
}

// ---- prefill attention on 32x32 MFMA tiles: 8 waves x 32 queries per workgroup -------------------------------------------------
// The same fragment algebra as attn_fwd_kernel on v_mfma_f32_32x32x16_bf16: S^T = K Q^T (a = K rows, b = Q), so a lane owns ONE
// query column (lane & 31) and 16 of the 32 keys of a sub-tile (the other 16 live in lane ^ 32): row maximum = in-lane chain + one
// cross-half exchange, row sum = in-lane partials merged once after the last tile.  The exponentiated scores, rounded to bf16 in
// pairs, ARE the B operand of O^T = V^T P^T — k-slot j of PV step t of sub-tile s is key 32 s + 16 t + 8 (j >> 2) + 4 hi + (j & 3), and
// the V^T A operand reads exactly those keys as two 8-byte pieces of its row (which is why V is cached transposed).
// Against the 16x16 form: half the LDS fragment bytes per flop, one cross-lane exchange per 64-key tile instead of four, K / V^T tiles
// staged once per 256 query-rows (QB queries x HPW heads of one KV head: with GQA the two query heads of a workgroup share the tile),
// a double-buffered LDS image with ONE barrier per tile, register prefetch two tiles ahead (issue-early / write-late), and output rows
// staged through LDS and stored as whole rows.
//   HPW = 1: 256 queries of one head per workgroup (ViT full attention, head dim 80 = 5 k-steps of 16, O^T padded to 96 rows);
//   HPW = 2: 128 queries x 2 query heads that share a KV head (LLM prefill, 16 q / 2 kv heads).
// Items: as attn_fwd_kernel (<= QB queries of one segment, a key range, optionally a second key range walked first).
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int HD, int HPW>
__global__ __launch_bounds__(512, 2) void attn_fwd32_kernel(const AttnParams p) {
    constexpr int WPH = 8 / HPW;              // waves per head
    constexpr int KB = 64;                    // keys per tile
    constexpr int NKC = HD / 16;              // k-steps of QK^T
    constexpr int NDB = (HD + 31) / 32;       // 32-row d blocks of O^T
    constexpr int HDV = NDB * 32;             // V^T rows in the LDS image (rows >= HD stay zero)
    constexpr int LDK = HD + 8;               // K row pitch (elements): 16-lane b128 groups hit 16 distinct 16-B bank groups
    constexpr int LDV = KB + 8;               // V^T row pitch: 144 B = 9 x 16 B (odd): the b128 lane groups hit 16 distinct 16-B bank groups
    constexpr int SK = KB * LDK, SV = HDV * LDV;
    constexpr int KCH = HD / 8;               // 16-B chunks per K row
    constexpr int NKR = (KB * KCH + 511) / 512;
    constexpr int NVR = (HD * (KB / 4) + 511) / 512;
    static_assert(HD % 16 == 0 && 8 % HPW == 0, "attn_fwd32: head dim / heads per workgroup");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem32[];
    uint16_t* sK = smem32;                    // [2][SK]
    uint16_t* sVT = smem32 + 2 * SK;          // [2][SV]

    // grid = (head groups, items): the dispatch order walks the ITEMS slowest, so a work list sorted by descending cost (ops.make_items: the
    // causal blocks with the most key tiles first) is longest-processing-time-first over the whole launch, and the workgroups that share an
    // item's K / V^T tiles start together
    const int item = blockIdx.y, hg = blockIdx.x;
    const AttnItem it = p.items[item];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 31, hi = lane >> 5;
    const int hsel = wave / WPH, wq = wave - hsel * WPH;
    const int h = hg * HPW + hsel;
    const int kvh = (hg * HPW) / p.group;                  // both heads of the workgroup share it (group % HPW == 0)
    const int wq0 = it.q_start + wq * 32;                  // this wave's first query
    const int q_idx = wq0 + qc;
    const bool q_ok = q_idx < it.q_end;
    const bool wave_on = wq0 < it.q_end;

    if constexpr (HDV > HD) {      // O^T pad rows of both buffers: zeroed once, never rewritten
        constexpr int PADW = (HDV - HD) * LDV / 2;     // dwords per buffer
        for (int q = tid; q < 2 * PADW; q += 512) {
            const int b = q >= PADW ? 1 : 0;
            reinterpret_cast<uint32_t*>(sVT + b * SV + HD * LDV)[q - b * PADW] = 0u;
        }
    }

    // Q fragments (B operand): lane (query qc, k-half hi) holds d = kc*16 + hi*8 .. +8
    bf16x8 qf[NKC];
    {
        const int q_ld = q_ok ? q_idx : it.q_end - 1;
        const uint16_t* qp = p.Q + (long long)q_ld * p.q_tok + (long long)h * p.q_head + hi * 8;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const uint4 v = *reinterpret_cast<const uint4*>(qp + kc * 16);
            qf[kc] = *reinterpret_cast<const bf16x8*>(&v);
        }
    }
    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;       // m_run: running maximum of c1 * score (base 2); l_run: THIS lane's share of the row sum
    const float c1 = p.scale * 1.44269504088896f;

    int own_hi = it.kv_end;
    if (p.causal && it.q_end < own_hi) own_hi = it.q_end;  // keys beyond the last query are never attended
    const uint16_t* Kb = p.K + (long long)kvh * p.k_head;
    const uint16_t* VTb = p.VT + (long long)kvh * HD * p.vt_row;

    // ---- staging: global -> registers two tiles ahead, registers -> LDS one tile ahead.  Raw buffer loads: a wave-uniform descriptor +
    // scalar tile offset + ONE loop-invariant 32-bit lane offset per piece (no address arithmetic in the loop, no load inside a branch).
    // The K descriptor ends at the range's last row, so the rows past it read as zero; V^T columns past the range are zeroed on the way
    // to LDS (last tile only: 0 x NaN would poison the PV product). ----
    typedef __attribute__((ext_vector_type(4))) unsigned int v4u;
    typedef __attribute__((ext_vector_type(2))) unsigned int v2u;
    const int k_tok = (int)p.k_tok, vt_row = (int)p.vt_row;
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)VTb, 0, (uint32_t)HD * (uint32_t)vt_row * 2u, 0x00020000);
    uint32_t koff[NKR], voff[NVR];
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
        int q = tid + i * 512;
        if (NKR * 512 != KB * KCH && q >= KB * KCH) q = KB * KCH - 1;
        const int r = q / KCH, c = q - r * KCH;
        koff[i] = (uint32_t)(r * k_tok + c * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
        int q = tid + i * 512;
        if (NVR * 512 != HD * (KB / 4) && q >= HD * (KB / 4)) q = HD * (KB / 4) - 1;
        voff[i] = (uint32_t)((q >> 4) * vt_row + (q & 15) * 4) * 2u;
    }
    v4u rk[NKR];
    v2u rv[NVR];
    auto gload = [&](int k0, int hi_) {
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (uint32_t)hi_ * (uint32_t)k_tok * 2u, 0x00020000);   // unsigned: rows up to 4 GiB from the base (host-checked)
#pragma unroll
        for (int i = 0; i < NKR; ++i) rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, koff[i], (int)((uint32_t)k0 * (uint32_t)k_tok * 2u), 0);
#pragma unroll
        for (int i = 0; i < NVR; ++i) rv[i] = __builtin_amdgcn_raw_buffer_load_b64(rsV, voff[i], k0 * 2, 0);
    };
    auto swrite = [&](int buf, int k0, int hi_) {
        const bool last = k0 + KB > hi_;       // wave-uniform: only a range's last tile has columns to zero
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = tid + i * 512;
            const int r = q / KCH, c = q - r * KCH;
            if (NKR * 512 == KB * KCH || q < KB * KCH) *reinterpret_cast<v4u*>(&sK[buf * SK + r * LDK + c * 8]) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = tid + i * 512;
            const int d = q >> 4, c = q & 15;
            v2u v = rv[i];
            if (last) {
                const int nb = min(max(hi_ - (k0 + c * 4), 0), 4) * 16;      // valid bits of this 4-key piece
                const unsigned long long m = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
                v.x &= (uint32_t)m;
                v.y &= (uint32_t)(m >> 32);
            }
            // 4-key pieces land with the low two bits of their index swapped: the 8 k-slots of a PV step (keys 16 t + 4 hi + {0..3, 8..11})
            // are then 16 contiguous bytes of the row — one ds_read_b128 per fragment
            const int cp = (c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1);
            if (NVR * 512 == HD * (KB / 4) || q < HD * (KB / 4)) *reinterpret_cast<v2u*>(&sVT[buf * SV + d * LDV + cp * 4]) = v;
        }
    };

    // one 64-key tile for this wave's 32 queries
    auto compute = [&](int buf, int k0, int hi_) {
        const uint16_t* bK = sK + buf * SK + qc * LDK + hi * 8;
        const uint16_t* bV = sVT + buf * SV + qc * LDV + hi * 8;
        // ---- S^T = K Q^T: every fragment read in front of the MFMAs (the two sub-tiles' accumulators alternate) ----
        bf16x8 kf[2][NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) kf[sub][kc] = *reinterpret_cast<const bf16x8*>(bK + sub * 32 * LDK + kc * 16);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][kc], qf[kc], s[sub], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments of the first 32 keys: in flight while the softmax runs
        bf16x8 va[2][NDB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < NDB; ++db) va[t][db] = *reinterpret_cast<const bf16x8*>(bV + db * 32 * LDV + t * 16);
        __builtin_amdgcn_sched_barrier(0);
        // s[sub][r] = raw score(key = k0 + sub*32 + (r&3) + 8*(r>>2) + 4*hi, query = q_idx); the mask only where a tile needs one (wave-uniform)
        const bool need_mask = (k0 + KB > hi_) || (p.causal && k0 + KB - 1 > wq0);
        if (need_mask) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < hi_ && (!p.causal || key <= q_idx);
                    s[sub][r] = ok ? s[sub][r] : -INFINITY;
                }
        }
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c1);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;       // no valid key yet: everything stays zero
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        // probabilities of one 32-key sub-tile -> the two B operands of its PV steps (k-slot j of step t = score register 8 t + j) + the lane's row-sum share
        auto probs = [&](const f32x16& sc, bf16x8 (&pf)[2]) -> float {
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t * 8 + 2 * j], c1, -m_use));       // masked: exp2(-inf) = 0
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t * 8 + 2 * j + 1], c1, -m_use));
                    ps += e0 + e1;
                    w[j] = pack_bf16x2(e0, e1);
                }
                const uint4 pk = uint4{w[0], w[1], w[2], w[3]};
                pf[t] = *reinterpret_cast<const bf16x8*>(&pk);
            }
            return ps;
        };
        bf16x8 pf0[2], pf1[2];
        const float ps0 = probs(s[0], pf0);
        if (__any(alpha != 1.0f)) {               // wave-uniform: later tiles rarely move the maximum
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        // ---- O^T = alpha O^T + V^T P^T.  The second 32 keys' fragments are requested first; the first 32 keys' MFMAs then run with the
        // second sub-tile's exponentials in their shadow (one MFMA : ~7 VALU issues), the second 32 keys' MFMAs close the tile. ----
        bf16x8 vb[2][NDB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < NDB; ++db) vb[t][db] = *reinterpret_cast<const bf16x8*>(bV + db * 32 * LDV + 32 + t * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[t][db], pf0[t], o[db], 0, 0, 0);
        const float ps1 = probs(s[1], pf1);
#pragma unroll
        for (int i = 0; i < 2 * NDB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, (56 + 2 * NDB - 1) / (2 * NDB), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        l_run = l_run * alpha + (ps0 + ps1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[t][db], pf1[t], o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- tile walk: the optional second range first (the shared prefix), then the item's own range ----
    int2 r2 = int2{0, 0};
    if (p.items2) r2 = p.items2[item];
    const bool has2 = r2.y > r2.x;
    int k0 = has2 ? r2.x : it.kv_start, khi = has2 ? r2.y : own_hi;
    bool in2 = has2;
    // (k1, khi1, in21) = the tile after (k0, khi, in2); (k2, ...) the one after that
    auto advance = [&](int& k, int& hi_, bool& in) {
        k += KB;
        if (k >= hi_ && in) { k = it.kv_start; hi_ = own_hi; in = false; }
    };
    int k1 = k0, khi1 = khi; bool in21 = in2;
    if (k0 < khi) gload(k0, khi);               // the first tile's loads go out right behind the Q loads: one round trip, not two
    // the Q fragments are complete HERE (the empty asm uses them; in-order returns: it waits for them, not for the newer tile loads): left
    // pending, hipcc's waitcnt pass puts `s_waitcnt vmcnt(7) .. vmcnt(0)` in front of the QK^T MFMAs inside the tile loop (the first trip
    // needs them), and every trip then drains the tile prefetch it has just issued
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) asm volatile("" : "+v"(qf[kc]));
    if (k0 < khi) {
        advance(k1, khi1, in21);
        swrite(0, k0, khi);
        if (k1 < khi1) gload(k1, khi1);
    }
    int buf = 0;
    while (k0 < khi) {
        __syncthreads();                        // tile (k0) is visible in sbuf[buf]; everyone is done with sbuf[buf ^ 1]
        int k2 = k1, khi2 = khi1; bool in22 = in21;
        if (k1 < khi1) {
            swrite(buf ^ 1, k1, khi1);
            advance(k2, khi2, in22);
            if (k2 < khi2) gload(k2, khi2);
        }
        // causal: a tile whose first key lies beyond this wave's last query is skipped by the wave (own range only)
        if (wave_on && !(p.causal && !in2 && k0 > wq0 + 31)) compute(buf, k0, khi);
        k0 = k1; khi = khi1; in2 = in21;
        k1 = k2; khi1 = khi2; in21 = in22;
        buf ^= 1;
    }

    // ---- output: normalise, stage the wave's 32 x HD block in its own slice of the (now free) LDS image, store whole rows ----
    constexpr int LDO = HD + 8;
    static_assert(8 * 32 * LDO <= 2 * (SK + SV), "attn_fwd32: output staging does not fit the LDS image");
    __syncthreads();
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    uint16_t* sO = smem32 + wave * 32 * LDO;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = db * 32 + g4 * 8 + hi * 4;
            if (HDV == HD || d < HD) {
                uint2 w;
                w.x = pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv);
                w.y = pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(sO + qc * LDO + d) = w;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // wave-private slice: program order, the fence pins hipcc
#pragma unroll
    for (int i = 0; i < (32 * KCH + 63) / 64; ++i) {
        const int q = lane + i * 64;
        const int r = q / KCH, c = q - r * KCH;
        if (q < 32 * KCH && wq0 + r < it.q_end) {
            const uint4 v = *reinterpret_cast<const uint4*>(sO + r * LDO + c * 8);
            *reinterpret_cast<uint4*>(p.O + (long long)(wq0 + r) * p.o_tok + (long long)h * p.o_head + c * 8) = v;
        }
    }
}

// out[head, d] = sum_s exp(m_s - M) O_s[d] / sum_s exp(m_s - M) l_s over the valid KV chunks (fixed order)
template <int HD>
__global__ __launch_bounds__(HD) void attn_decode_combine_kernel(const float* __restrict__ part, const int* __restrict__ dyn_kv_len,
                                                                 int kv_chunk, int n_kv_heads, int group, uint16_t* __restrict__ out,
                                                                 const int* __restrict__ seq_state, long long part_seq_stride, long long out_seq_stride) {
    const int head = blockIdx.x, d = threadIdx.x;
    const int kvh = head / group, slot = head - kvh * group;
    int n_keys;
    if (seq_state) {   // sequence blockIdx.y of a decode batch
        const int* st = seq_state + blockIdx.y * 8;
        out += (long long)blockIdx.y * out_seq_stride;
        if (st[3]) {                    // finished: the split kernel skipped it; the row is zeroed (it must be finite, see attn_fwd_kernel)
            out[(long long)head * HD + d] = 0;
            return;
        }
        n_keys = st[0] + 1 - st[2];
        part += (long long)blockIdx.y * part_seq_stride;
    } else {
        n_keys = *dyn_kv_len;
    }
    const int n_valid = (n_keys + kv_chunk - 1) / kv_chunk;
    // the arithmetic lives in attn_combine_row (decode_common.h): the decode GEMV's fused prologue (M <= 2) runs the same code, so a sequence's
    // attention rows are the same bits with and without this launch
    float o1[1];
    attn_combine_row<1>(part + ((long long)kvh * 16 + slot) * (HD + 2), (long long)n_kv_heads * 16 * (HD + 2), n_valid, d, o1);
    out[(long long)head * HD + d] = f32_to_bf16(o1[0]);
}

// ---- decode attention, one workgroup per (KV head, sequence[, KV split]) ---------------------------------------------------
// The query heads that share a KV head ride as the MFMA columns (GQA: 8 of 16 columns at 16q/2kv).  The NW waves of a workgroup
// take the 64-key tiles of the range round-robin; every tile goes straight from the cache to MFMA fragments in registers (K rows
// as the A operand of S^T = K Q^T, V^T pieces as the A operand of O^T = V^T P^T — the same fragment algebra as attn_fwd_kernel,
// minus the LDS staging: nothing is shared between waves), all 48 loads of a tile in flight at once.  The waves' (O, m, l) meet
// in LDS and are merged in wave order; with one split the normalised bf16 rows are written directly (ONE launch per layer
// instead of split + combine), with several the merged partial of each split goes to `part` for attn_decode_combine_kernel.
// MEASURED (MI355X, 651-key contexts, gpurun_out/r02_run17): 16.8 us per layer at every batch size against 5.3 + 4.3 us (B = 1) ..
// 13.0 + 4.5 us (B = 16) for the 64-key split kernel + combine: all of a (KV head, sequence)'s K/V^T (333 KB) funnels through
// ONE CU in fragment-shaped 64-byte / 8-byte pieces (20 GB/s), where the split kernel spreads 64-key chunks over 22+ CUs and
// stages them coalesced through LDS.  Kept as an A/B option (fo1_attention_decode_set_impl), not the default.
#ifdef FO1_ENABLE_AB      // one workgroup per (KV head, sequence): measured slower than split + combine, A/B only
struct AttnDecParams {
    const uint16_t* Q; long long q_seq_stride;       // q rows [B][n_q_heads * 128]
    const uint16_t* K; long long k_tok, k_head;
    const uint16_t* VT; long long vt_row;
    uint16_t* O; long long o_seq_stride;
    float* part; long long part_seq_stride;          // [split][kv head][16][HD + 2] per sequence
    const int* seq_state;                            // [B][8] (decode.hip) or null
    const int* dyn_kv_len;                           // single-sequence form: keys [0, *dyn_kv_len)
    int n_kv_heads, group, split_keys, n_splits;
    float scale;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_wg_kernel(const AttnDecParams p) {
    constexpr int HD = 128, NC = 4, NDB = 8, PITCH = HD + 4;
    __shared__ __attribute__((aligned(16))) float sp[NW * 16 * PITCH];
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, g = lane >> 4;
    int kv0 = 0, kv_len;
    if (p.seq_state) {
        const int* st = p.seq_state + b * 8;
        kv0 = st[2];
        kv_len = st[0] + 1;
    } else {
        kv_len = *p.dyn_kv_len;
    }
    const int lo = kv0 + split * p.split_keys;
    const int hi = min(kv_len, lo + p.split_keys);
    if (lo >= hi) return;                            // (several splits only) the combine pass skips empty splits

    // Q fragments (B operand): lane (query slot ql, k-group g) holds d = c*32 + g*8 .. +8; slots >= group are zero columns
    bf16x8 qf[NC];
    {
        const uint16_t* qp = p.Q + (long long)b * p.q_seq_stride + (long long)(kvh * p.group + (ql < p.group ? ql : 0)) * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint4 v = *reinterpret_cast<const uint4*>(qp + c * 32 + g * 8);
            if (ql >= p.group) v = uint4{0, 0, 0, 0};
            qf[c] = *reinterpret_cast<bf16x8*>(&v);
        }
    }
    f32x4 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const uint16_t* Kb = p.K + (long long)kvh * p.k_head;
    const uint16_t* VTb = p.VT + (long long)kvh * HD * p.vt_row;

    for (int k0 = lo + wave * 64; k0 < hi; k0 += NW * 64) {
        // ---- every load of the tile first: K rows (fragment shape: 16 keys x 64 B per instruction), V^T 8-byte pieces ----
        uint4 kf[4][NC];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = k0 + kt * 16 + ql;
            const uint16_t* kp = Kb + (long long)(key < hi ? key : hi - 1) * p.k_tok + g * 8;
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[kt][c] = *reinterpret_cast<const uint4*>(kp + c * 32);
        }
        uint2 vlo[NDB][2], vhi[NDB][2];
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const uint16_t* vr = VTb + (long long)(db * 16 + ql) * p.vt_row;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // a-operand k-slots: j<4 -> key (2*half)*16 + g*4 + j ; j>=4 -> key (2*half+1)*16 + g*4 + (j-4)
                const int ka = k0 + half * 32 + g * 4, kb2 = ka + 16;
                vlo[db][half] = *reinterpret_cast<const uint2*>(vr + (ka < hi ? ka : lo));
                vhi[db][half] = *reinterpret_cast<const uint2*>(vr + (kb2 < hi ? kb2 : lo));
            }
        }
        // ---- S^T = K Q^T ----
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c)
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&kf[kt][c]), qf[c], s[kt], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + g * 4 + r;
                const float v = key < hi ? s[kt][r] * p.scale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.0f : __expf(m_run - m_new);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t w[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = half * 2 + t;
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = (s[kt][r] == -INFINITY) ? 0.f : __expf(s[kt][r] - m_new);
                    e[r] = bf16_to_f32(f32_to_bf16(e[r]));      // sum what is multiplied into V: the bf16-rounded probabilities
                    psum += e[r];
                }
                w[t * 2 + 0] = pack_bf16x2(e[0], e[1]);
                w[t * 2 + 1] = pack_bf16x2(e[2], e[3]);
            }
            uint4 pk = uint4{w[0], w[1], w[2], w[3]};
            pf[half] = *reinterpret_cast<bf16x8*>(&pk);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // ---- O^T = alpha * O^T + V^T P^T (keys past the range carry p = 0 against finite cache contents) ----
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint4 vk = uint4{vlo[db][half].x, vlo[db][half].y, vhi[db][half].x, vhi[db][half].y};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&vk), pf[half], o[db], 0, 0, 0);
            }
        }
    }
    // ---- the waves' (O, m, l) of query slot ql meet in LDS; o[db][r] = O[slot ql][d = db*16 + g*4 + r] ----
    {
        float* pr = sp + (wave * 16 + ql) * PITCH;
#pragma unroll
        for (int db = 0; db < NDB; ++db) *reinterpret_cast<float4*>(pr + db * 16 + g * 4) = float4{o[db][0], o[db][1], o[db][2], o[db][3]};
        if (g == 0) { pr[HD] = m_run; pr[HD + 1] = l_run; }
    }
    __syncthreads();
    for (int t = tid; t < p.group * HD; t += NW * 64) {
        const int slot = t >> 7, d = t & (HD - 1);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sp[(w * 16 + slot) * PITCH + HD]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = sp[(w * 16 + slot) * PITCH + HD];
            const float wgt = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            num += wgt * sp[(w * 16 + slot) * PITCH + d];
            den += wgt * sp[(w * 16 + slot) * PITCH + HD + 1];
        }
        if (p.n_splits == 1) {
            p.O[(long long)b * p.o_seq_stride + (long long)(kvh * p.group + slot) * HD + d] = f32_to_bf16(den > 0.f ? num / den : 0.f);
        } else {
            float* pr = p.part + (long long)b * p.part_seq_stride + (((long long)split * p.n_kv_heads + kvh) * 16 + slot) * (HD + 2);
            pr[d] = num;
            if (d == 0) { pr[HD] = M; pr[HD + 1] = den; }
        }
    }
}

static int g_attn_decode_impl = 0;   // 0 = 64-key split-KV partials + combine; 1 = attn_decode_wg_kernel (measured slower: see the note below)

// One launch for slots up to 2048 rows (32 tiles over 8 waves); longer slots: 1024-key splits + the fixed-order combine.
static int launch_attn_decode_wg(AttnDecParams& p, int max_kv_len, int batch, int n_q_heads, hipStream_t st) {
    constexpr int NW = 8;
    if (max_kv_len <= 2048) { p.n_splits = 1; p.split_keys = cdiv(max_kv_len, 64) * 64; }
    else { p.split_keys = 1024; p.n_splits = cdiv(max_kv_len, 1024); }
    p.part_seq_stride = (long long)p.n_splits * p.n_kv_heads * 16 * (128 + 2);
    FO1_LAUNCH("attn_decode_wg", (double)batch * max_kv_len * p.n_kv_heads * 128 * 4.0, (attn_decode_wg_kernel<NW>),
               dim3(p.n_splits, p.n_kv_heads, batch), dim3(NW * 64), 0, st, p);
    if (p.n_splits > 1)
        FO1_LAUNCH("attn_decode_combine", (double)batch * n_q_heads * 128 * 8.0, attn_decode_combine_kernel<128>, dim3(n_q_heads, batch), dim3(128), 0,
                   st, (const float*)p.part, p.dyn_kv_len, p.split_keys, p.n_kv_heads, p.group, p.O, p.seq_state, p.part_seq_stride, p.o_seq_stride);
    return FO1_OK;
}

// impl 2 (round 4, A/B): one workgroup per (KV head, sequence) whose FOUR WAVES EACH WALK THEIR OWN 64-key tiles (wave w: tiles w, w + 4, ...)
// through a wave-private LDS image — the per-tile arithmetic, layouts and the coalesced 16-B / 8-B loads prefetched one tile ahead in registers are
// attn_fwd_kernel's, but nothing in the loop waits for another wave: the dependent chain of a 700-key context is 3 tiles instead of 11.  The four
// (m, l, O) meet in LDS once, merged in wave order.  Batched decode only (seq_state); any context length.
__global__ __launch_bounds__(256) void attn_decode_ws_kernel(const AttnDecParams p) {
    constexpr int HD = 128, NC = 4, NDB = 8, KB = 64, LDKR = HD + 8, LDVT = KB + 4, PITCH = HD + 2;
    constexpr int WAVE_LDS = (KB * LDKR + HD * LDVT) * 2;          // bytes per wave: K tile + V^T tile
    extern __shared__ __attribute__((aligned(16))) char smem_ws[];
    const int kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int* st = p.seq_state + b * 8;
    if (st[3]) return;                               // finished / empty slot (workgroup-uniform)
    const int kv0 = st[2], kv_len = st[0] + 1;
    uint16_t* sK = reinterpret_cast<uint16_t*>(smem_ws + wave * WAVE_LDS);
    uint16_t* sVT = sK + KB * LDKR;

    bf16x8 qf[NC];
    {
        const uint16_t* qp = p.Q + (long long)b * p.q_seq_stride + ((long long)kvh * p.group + (ql < p.group ? ql : 0)) * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint4 v = uint4{0, 0, 0, 0};
            if (ql < p.group) v = *reinterpret_cast<const uint4*>(qp + c * 32 + g * 8);
            qf[c] = *reinterpret_cast<bf16x8*>(&v);
        }
    }
    f32x4 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float c1 = p.scale * 1.44269504088896f;
    const uint16_t* Kb = p.K + (long long)kvh * p.k_head;
    const uint16_t* VTb = p.VT + (long long)kvh * HD * p.vt_row;

    constexpr int NKR = KB * (HD / 8) / 64;          // 16-B K chunks per lane per tile (16)
    constexpr int NVR = HD * (KB / 4) / 64;          // 8-B V^T pieces per lane per tile (32)
    uint4 rk[NKR];
    uint2 rv[NVR];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = lane + i * 64, r = q >> 4, c = q & 15;
            rk[i] = uint4{0, 0, 0, 0};
            if (k0 + r < kv_len) rk[i] = *reinterpret_cast<const uint4*>(Kb + (long long)(k0 + r) * p.k_tok + c * 8);
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = lane + i * 64, d = q >> 4, c = q & 15;
            rv[i] = uint2{0, 0};
            if (k0 + c * 4 < kv_len) rv[i] = *reinterpret_cast<const uint2*>(VTb + (long long)d * p.vt_row + k0 + c * 4);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NKR; ++i) {
            const int q = lane + i * 64, r = q >> 4, c = q & 15;
            *reinterpret_cast<uint4*>(&sK[r * LDKR + c * 8]) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int q = lane + i * 64, d = q >> 4, c = q & 15;
            *reinterpret_cast<uint2*>(&sVT[d * LDVT + c * 4]) = rv[i];
        }
    };
    int k0 = kv0 + wave * KB;
    if (k0 < kv_len) gload(k0);
    for (; k0 < kv_len; k0 += 4 * KB) {
        // wave-private image: this wave's LDS instructions execute in order, so the tile's writes follow the previous tile's reads and
        // precede this tile's; the fences only keep hipcc from moving them
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        swrite();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int nk0 = k0 + 4 * KB;
        if (nk0 < kv_len) gload(nk0);

        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8 kf[2][NC];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[kt][c] = *reinterpret_cast<const bf16x8*>(&sK[((kh * 2 + kt) * 16 + ql) * LDKR + c * 32 + g * 8]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    s[kh * 2 + kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[c], s[kh * 2 + kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint2 vlo[NDB], vhi[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + g * 4];
            vlo[db] = *reinterpret_cast<const uint2*>(vr);
            vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + KB > kv_len) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + kt * 16 + g * 4 + r >= kv_len) s[kt][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3])), fmaxf(fmaxf(s[3][0], s[3][1]), fmaxf(s[3][2], s[3][3]))));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c1);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t w[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = half * 2 + t;
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c1, -m_use));
                    psum += e[r];
                }
                w[t * 2 + 0] = pack_bf16x2(e[0], e[1]);
                w[t * 2 + 1] = pack_bf16x2(e[2], e[3]);
            }
            uint4 pk = uint4{w[0], w[1], w[2], w[3]};
            pf[half] = *reinterpret_cast<bf16x8*>(&pk);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < NDB; ++db) { o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha; }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 vk[NDB];
#pragma unroll
            for (int db = 0; db < NDB; ++db) vk[db] = uint4{vlo[db].x, vlo[db].y, vhi[db].x, vhi[db].y};
            if (half == 0) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const uint16_t* vr = &sVT[(db * 16 + ql) * LDVT + 32 + g * 4];
                    vlo[db] = *reinterpret_cast<const uint2*>(vr);
                    vhi[db] = *reinterpret_cast<const uint2*>(vr + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&vk[db]), pf[half], o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the waves' partials meet in LDS (each wave re-uses its own image): [16 queries][HD + 2] fp32 = O row, m (base 2), l
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    float* sp = reinterpret_cast<float*>(smem_ws + wave * WAVE_LDS);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        *reinterpret_cast<float4*>(sp + ql * PITCH + db * 16 + g * 4) = float4{o[db][0], o[db][1], o[db][2], o[db][3]};
    if (g == 0) { sp[ql * PITCH + HD] = m_run; sp[ql * PITCH + HD + 1] = l_run; }
    __syncthreads();
    const int d = tid & (HD - 1);
    for (int qq = tid >> 7; qq < p.group; qq += 2) {
        float mw[4], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = reinterpret_cast<const float*>(smem_ws + w * WAVE_LDS)[qq * PITCH + HD];
            M = fmaxf(M, mw[w]);
        }
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* pw = reinterpret_cast<const float*>(smem_ws + w * WAVE_LDS) + qq * PITCH;
            const float f = (mw[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[w] - M);
            num += f * pw[d];
            den += f * pw[HD + 1];
        }
        p.O[(long long)b * p.o_seq_stride + ((long long)kvh * p.group + qq) * HD + d] = f32_to_bf16(den > 0.f ? num / den : 0.f);
    }
}

static int launch_attn_decode_ws(AttnDecParams& p, int max_kv_len, int batch, hipStream_t st) {
    constexpr int smem = 4 * (64 * (128 + 8) + 128 * (64 + 4)) * 2;
    static bool attr_done = false;
    if (!attr_done) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)attn_decode_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    FO1_LAUNCH("attn_decode_ws", (double)batch * max_kv_len * p.n_kv_heads * 128 * 4.0, attn_decode_ws_kernel, dim3(1, p.n_kv_heads, batch), dim3(256), smem, st, p);
    return FO1_OK;
}

#endif   // FO1_ENABLE_AB (attn_decode_wg_kernel)

extern int g_gemv_profile_shapes;   // gemv.hip: per-shape profile rows (fo1_gemm_profile_shapes)

template <int HD, int HPW>
static int launch_attn32(const AttnParams& p, hipStream_t st, double flops) {
    constexpr int LDK = HD + 8, LDV = 64 + 8, HDV = (HD + 31) / 32 * 32;
    constexpr int smem = 2 * (64 * LDK + HDV * LDV) * 2;
    static bool attr_done = false;      // > 64 KB of dynamic LDS needs the attribute (first call of an instantiation; never inside a capture: passes run eagerly first)
    if (!attr_done) {
        FO1_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd32_kernel<HD, HPW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    char pname[48];
    const char* name = "attn_fwd32";
    if (profile_enabled() && g_gemv_profile_shapes) {
        snprintf(pname, sizeof pname, "attn_fwd32 hd%d q%d items%d heads%d%s", HD, 256 / HPW, p.n_items, p.Hq, p.causal ? " causal" : "");
        name = pname;
    }
    FO1_LAUNCH(name, flops, (attn_fwd32_kernel<HD, HPW>), dim3(p.Hq / HPW, p.n_items), dim3(512), smem, st, p);
    return FO1_OK;
}

template <int HD>
static int launch_attn(const AttnParams& p, int q_block, hipStream_t st, double flops) {
    char pname[48];
    const char* name = "attn_fwd";
    if (profile_enabled() && g_gemv_profile_shapes) {
        snprintf(pname, sizeof pname, "attn_fwd hd%d q%d items%d heads%d%s", HD, q_block, p.n_items, p.Hq, p.causal ? " causal" : "");
        name = pname;
    }
    if (q_block == 16)
        FO1_LAUNCH(name, flops, (attn_fwd_kernel<HD, 1>), dim3(p.n_items, p.Hq), dim3(64), 0, st, p);
    else if (q_block == 32)
        FO1_LAUNCH(name, flops, (attn_fwd_kernel<HD, 2>), dim3(p.n_items, p.Hq), dim3(128), 0, st, p);
    else
        FO1_LAUNCH(name, flops, (attn_fwd_kernel<HD, 4>), dim3(p.n_items, p.Hq), dim3(256), 0, st, p);
    return FO1_OK;
}

}  // namespace fo1

extern "C" {

static int attention_entry(const void* Q, long long q_tok_stride, long long q_head_stride,
                       const void* K, long long k_tok_stride, long long k_head_stride,
                       const void* VT, long long vt_row_stride,
                       void* O, long long o_tok_stride, long long o_head_stride,
                       const int32_t* items, int n_items, int q_block, int n_q_heads, int n_kv_heads, int head_dim,
                       float scale, int causal, const int32_t* q_row_base, double flops_hint, void* stream,
                       const float* bias, int wlen, int sw_ws, int sw_shift, int sw_nwy, int sw_nwx, const int32_t* prefix_ranges = nullptr) {
    using namespace fo1;
    if (n_items == 0) return FO1_OK;
    FO1_CHECK_ARG(Q && K && VT && O && items, "attention: NULL operand");
    FO1_CHECK_ARG(n_items > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, "attention: bad head counts");
    FO1_CHECK_ARG(head_dim == 32 || head_dim == 80 || head_dim == 128, "attention: head_dim %d not built (32, 80, 128)", head_dim);
    FO1_CHECK_ARG(q_block == 16 || q_block == 32 || q_block == 64 || q_block == 128 || q_block == 256,
                  "attention: q_block %d must be 16, 32, 64 (16x16 MFMA form) or 128, 256 (32x32 form)", q_block);
    FO1_CHECK_ARG(q_tok_stride % 8 == 0 && q_head_stride % 8 == 0 && k_tok_stride % 8 == 0 && k_head_stride % 8 == 0,
                  "attention: Q/K strides must be multiples of 8 elements");
    FO1_CHECK_ARG(vt_row_stride % 4 == 0 && o_tok_stride % 4 == 0 && o_head_stride % 4 == 0,
                  "attention: V^T / O strides must be multiples of 4 elements");
    FO1_CHECK_ARG(((uintptr_t)Q & 15) == 0 && ((uintptr_t)K & 15) == 0 && ((uintptr_t)VT & 7) == 0 && ((uintptr_t)O & 7) == 0,
                  "attention: misaligned operand");
    AttnParams p;
    p.Q = (const uint16_t*)Q; p.q_tok = q_tok_stride; p.q_head = q_head_stride;
    p.K = (const uint16_t*)K; p.k_tok = k_tok_stride; p.k_head = k_head_stride;
    p.VT = (const uint16_t*)VT; p.vt_row = vt_row_stride;
    p.O = (uint16_t*)O; p.o_tok = o_tok_stride; p.o_head = o_head_stride;
    p.items = (const AttnItem*)items;
    p.items2 = (const int2*)prefix_ranges;
    p.n_items = n_items; p.Hq = n_q_heads; p.group = n_q_heads / n_kv_heads;
    p.scale = scale; p.causal = causal; p.q_row_base = (const int*)q_row_base;
    p.part = nullptr; p.dyn_kv_len = nullptr; p.kv_chunk = 0; p.q_range_end = 0; p.part_tiles = 0; p.grid_batch = 0;
    p.seq_state = nullptr; p.q_seq_stride = 0; p.part_seq_stride = 0;
    p.bias = bias; p.wlen = wlen; p.sw_ws = sw_ws; p.sw_shift = sw_shift; p.sw_nwy = sw_nwy; p.sw_nwx = sw_nwx;
    hipStream_t st = (hipStream_t)stream;
    if (q_block >= 128) {
        // 32x32-MFMA form (attn_fwd32_kernel): 256 queries of one head, or 128 queries x the 2 query heads of one KV head, per workgroup
        const int hpw = 256 / q_block;
        FO1_CHECK_ARG(head_dim == 80 || head_dim == 128, "attention: q_block %d is built for head_dim 80 and 128 (got %d)", q_block, head_dim);
        FO1_CHECK_ARG(!bias && !q_row_base, "attention: q_block %d takes no bias operand and no q_row_base", q_block);
        FO1_CHECK_ARG(n_items <= 65535, "attention: q_block %d walks the items on grid.y: at most 65535 (got %d)", q_block, n_items);
        FO1_CHECK_ARG(n_q_heads % hpw == 0 && (n_q_heads / n_kv_heads) % hpw == 0,
                      "attention: q_block 128 needs an even number of query heads per KV head (%d / %d)", n_q_heads, n_kv_heads);
        FO1_CHECK_ARG(o_tok_stride % 8 == 0 && o_head_stride % 8 == 0 && ((uintptr_t)O & 15) == 0, "attention: q_block %d stores 16-byte pieces: O misaligned", q_block);
        FO1_CHECK_ARG(k_tok_stride < (1 << 22) && vt_row_stride < (1 << 22), "attention: q_block %d addresses a tile with 32-bit offsets: K / V^T row stride too large", q_block);
        if (head_dim == 80) return hpw == 2 ? launch_attn32<80, 2>(p, st, flops_hint) : launch_attn32<80, 1>(p, st, flops_hint);
        return hpw == 2 ? launch_attn32<128, 2>(p, st, flops_hint) : launch_attn32<128, 1>(p, st, flops_hint);
    }
    if (head_dim == 32) return launch_attn<32>(p, q_block, st, flops_hint);
    if (head_dim == 80) return launch_attn<80>(p, q_block, st, flops_hint);
    return launch_attn<128>(p, q_block, st, flops_hint);
}

int fo1_attention_bf16(const void* Q, long long q_tok_stride, long long q_head_stride,
                       const void* K, long long k_tok_stride, long long k_head_stride,
                       const void* VT, long long vt_row_stride,
                       void* O, long long o_tok_stride, long long o_head_stride,
                       const int32_t* items, int n_items, int q_block, int n_q_heads, int n_kv_heads, int head_dim,
                       float scale, int causal, const int32_t* q_row_base, double flops_hint, void* stream) {
    return attention_entry(Q, q_tok_stride, q_head_stride, K, k_tok_stride, k_head_stride, VT, vt_row_stride, O, o_tok_stride, o_head_stride, items, n_items,
                           q_block, n_q_heads, n_kv_heads, head_dim, scale, causal, q_row_base, flops_hint, stream, nullptr, 0, 0, 0, 0, 0);
}

// fo1_attention_bf16 with a second key range per item (prefix_ranges: int32 [n_items][2] = [start, end), empty when start >= end) that
// every query of the item attends in full, before its own (causal) range: several prompts over ONE image share the rows of their common
// prefix (system text + the image tokens) — the prefix rows run through the layer once, every prompt's remaining rows attend
// [prefix | own rows].  The prefix rows must precede the item's rows in the index space.  (The reference runs the whole model once per
// prompt, mm_utils.py:600 caps a prompt at 100 region features: BASELINE configs[4]'s 300 proposals are three such prompts.)
int fo1_attention_windows_bf16(const void* Q, long long q_tok_stride, long long q_head_stride, const void* K, long long k_tok_stride,
                                long long k_head_stride, const void* VT, long long vt_row_stride, void* O, long long o_tok_stride, long long o_head_stride,
                                long long o_rows, const int32_t* items, int n_items, int n_q_heads, int n_kv_heads, int head_dim, float scale,
                                double flops_hint, void* stream) {
    using namespace fo1;
    if (n_items == 0) return FO1_OK;
    FO1_CHECK_ARG(Q && K && VT && O && items, "attention_windows: NULL operand");
    FO1_CHECK_ARG(head_dim == 80, "attention_windows: built for head dim 80 (got %d)", head_dim);
    FO1_CHECK_ARG(n_items > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && n_q_heads <= 65535, "attention_windows: bad head / item counts");
    FO1_CHECK_ARG(q_tok_stride % 8 == 0 && q_head_stride % 8 == 0 && k_tok_stride % 8 == 0 && k_head_stride % 8 == 0 && vt_row_stride % 4 == 0 &&
                  o_tok_stride % 4 == 0 && o_head_stride % 4 == 0, "attention_windows: strides (Q / K: multiples of 8 elements, V^T / O: of 4)");
    FO1_CHECK_ARG(((uintptr_t)Q & 15) == 0 && ((uintptr_t)K & 15) == 0 && ((uintptr_t)VT & 7) == 0 && ((uintptr_t)O & 7) == 0, "attention_windows: misaligned operand");
    const long long o_bytes = (o_rows - 1) * o_tok_stride * 2 + ((long long)(n_q_heads - 1) * o_head_stride + head_dim) * 2;
    FO1_CHECK_ARG(o_rows > 0 && o_bytes <= (1ll << 31), "attention_windows: the output spans %lld bytes (32-bit store offsets: at most 2 GiB)", o_bytes);
    AttnParams p;
    p.Q = (const uint16_t*)Q; p.q_tok = q_tok_stride; p.q_head = q_head_stride;
    p.K = (const uint16_t*)K; p.k_tok = k_tok_stride; p.k_head = k_head_stride;
    p.VT = (const uint16_t*)VT; p.vt_row = vt_row_stride;
    p.O = (uint16_t*)O; p.o_tok = o_tok_stride; p.o_head = o_head_stride;
    p.items = (const AttnItem*)items; p.items2 = nullptr;
    p.n_items = n_items; p.Hq = n_q_heads; p.group = n_q_heads / n_kv_heads;
    p.scale = scale; p.causal = 0; p.q_row_base = nullptr;
    p.part = nullptr; p.dyn_kv_len = nullptr; p.kv_chunk = 0; p.q_range_end = 0; p.part_tiles = 0; p.grid_batch = 0;
    p.seq_state = nullptr; p.q_seq_stride = 0; p.part_seq_stride = 0;
    p.bias = nullptr; p.wlen = 0; p.sw_ws = 0; p.sw_shift = 0; p.sw_nwy = 0; p.sw_nwx = 0;
    constexpr int IPW = 4;
    FO1_LAUNCH("attn_win1", flops_hint, (attn_win1_kernel<80, IPW>), dim3(cdiv(n_items, IPW), n_q_heads), dim3(256), 0, (hipStream_t)stream, p, (uint32_t)o_bytes);
    return FO1_OK;
}

int fo1_attention_prefix_bf16(const void* Q, long long q_tok_stride, long long q_head_stride,
                              const void* K, long long k_tok_stride, long long k_head_stride,
                              const void* VT, long long vt_row_stride,
                              void* O, long long o_tok_stride, long long o_head_stride,
                              const int32_t* items, const int32_t* prefix_ranges, int n_items, int q_block, int n_q_heads, int n_kv_heads,
                              int head_dim, float scale, int causal, double flops_hint, void* stream) {
    FO1_CHECK_ARG(prefix_ranges != nullptr && ((uintptr_t)prefix_ranges & 7) == 0, "attention_prefix: prefix_ranges NULL or misaligned");
    return attention_entry(Q, q_tok_stride, q_head_stride, K, k_tok_stride, k_head_stride, VT, vt_row_stride, O, o_tok_stride, o_head_stride, items, n_items,
                           q_block, n_q_heads, n_kv_heads, head_dim, scale, causal, nullptr, flops_hint, stream, nullptr, 0, 0, 0, 0, 0, prefix_ranges);
}

// Window attention with an additive bias (Swin W-MSA / SW-MSA, backbone/swin.py:136-175): softmax(q k^T scale + bias[head][i][j]
// (+ -100 between tokens of different shift regions)) v.  Every item's kv range is one window of `wlen` tokens (<= 256), windows
// stored consecutively (image-major, row-major over the nwy x nwx windows of an image); bias fp32 [n_heads][wlen][wlen];
// shift > 0 turns the shifted-window mask on (regions of BasicLayer.forward :446-466, computed from the window's place).
int fo1_attention_window_bias_bf16(const void* Q, long long q_tok_stride, long long q_head_stride, const void* K, long long k_tok_stride,
                                   long long k_head_stride, const void* VT, long long vt_row_stride, void* O, long long o_tok_stride,
                                   long long o_head_stride, const int32_t* items, int n_items, int q_block, int n_heads, int head_dim, float scale,
                                   const float* bias, int wlen, int ws, int shift, int nwy, int nwx, double flops_hint, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(bias && wlen > 0 && wlen <= 256 && ws > 0 && ws * ws == wlen && shift >= 0 && shift < ws && nwy > 0 && nwx > 0,
                  "attention_window_bias: bad window geometry (wlen=%d ws=%d shift=%d)", wlen, ws, shift);
    return attention_entry(Q, q_tok_stride, q_head_stride, K, k_tok_stride, k_head_stride, VT, vt_row_stride, O, o_tok_stride, o_head_stride, items, n_items,
                           q_block, n_heads, n_heads, head_dim, scale, 0, nullptr, flops_hint, stream, bias, wlen, ws, shift, nwy, nwx);
}

// Decode-step attention for ONE new token against the KV cache, split over KV chunks of 64 keys
// ("flash-decoding"): grid = (max_chunks, n_kv_heads); the n_q_heads/n_kv_heads query heads that share a KV head
// ride as the query columns of the MFMA.  Chunk count follows the DEVICE-side kv length (state[0] + 1 keys), so the
// launch is graph-replayable for every step; partial (O, m, l) are merged in a fixed order by a second kernel.
size_t fo1_attention_decode_workspace_bytes(int max_kv_len, int n_kv_heads, int head_dim) {
    return (size_t)fo1::cdiv(max_kv_len, 64) * n_kv_heads * 16 * (head_dim + 2) * sizeof(float);
}

int fo1_attention_decode_bf16(const void* q, const void* kcache, long long k_tok_stride, long long k_head_stride, const void* vtcache,
                              long long vt_row_stride, void* out, const int32_t* dyn_kv_len, int max_kv_len, int n_q_heads,
                              int n_kv_heads, int head_dim, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(q && kcache && vtcache && out && dyn_kv_len && workspace, "attention_decode: NULL operand");
    FO1_CHECK_ARG(head_dim == 128, "attention_decode: head_dim %d not built (128)", head_dim);
    FO1_CHECK_ARG(n_q_heads % n_kv_heads == 0 && n_q_heads / n_kv_heads <= 16, "attention_decode: at most 16 query heads per KV head");
    FO1_CHECK_ARG(vt_row_stride % 4 == 0 && k_tok_stride % 8 == 0 && k_head_stride % 8 == 0, "attention_decode: bad strides");
    if (workspace_bytes < fo1_attention_decode_workspace_bytes(max_kv_len, n_kv_heads, head_dim))
        return set_err(FO1_ERR_WORKSPACE, "attention_decode: workspace too small");
    static const AttnItem* one_item = nullptr;
    const int group = n_q_heads / n_kv_heads;
#ifdef FO1_ENABLE_AB
    if (g_attn_decode_impl == 1) {
        AttnDecParams d;
        d.Q = (const uint16_t*)q; d.q_seq_stride = 0;
        d.K = (const uint16_t*)kcache; d.k_tok = k_tok_stride; d.k_head = k_head_stride;
        d.VT = (const uint16_t*)vtcache; d.vt_row = vt_row_stride;
        d.O = (uint16_t*)out; d.o_seq_stride = 0;
        d.part = (float*)workspace; d.seq_state = nullptr; d.dyn_kv_len = (const int*)dyn_kv_len;
        d.n_kv_heads = n_kv_heads; d.group = group; d.scale = scale;
        return launch_attn_decode_wg(d, max_kv_len, 1, n_q_heads, (hipStream_t)stream);
    }
#endif
    AttnParams p;
    p.Q = (const uint16_t*)q; p.q_tok = head_dim; p.q_head = (long long)group * head_dim;   // "queries" walk the heads of a group
    p.K = (const uint16_t*)kcache; p.k_tok = k_tok_stride; p.k_head = k_head_stride;
    p.VT = (const uint16_t*)vtcache; p.vt_row = vt_row_stride;
    p.O = nullptr; p.o_tok = 0; p.o_head = 0;
    p.items = (const AttnItem*)workspace;   // unused in PARTIAL mode except q range below (read from a device constant)
    p.n_items = cdiv(max_kv_len, 64); p.Hq = n_kv_heads; p.group = 1;
    p.scale = scale; p.causal = 0; p.q_row_base = nullptr;
    p.part = (float*)workspace; p.dyn_kv_len = (const int*)dyn_kv_len; p.kv_chunk = 64;
    (void)one_item;
    p.items = nullptr; p.items2 = nullptr;
    p.q_range_end = group; p.part_tiles = 0; p.grid_batch = 0;
    p.seq_state = nullptr; p.q_seq_stride = 0; p.part_seq_stride = 0;
    p.bias = nullptr; p.wlen = 0; p.sw_ws = p.sw_shift = p.sw_nwy = p.sw_nwx = 0;
    hipStream_t st = (hipStream_t)stream;
    FO1_LAUNCH("attn_decode_split", (double)max_kv_len * n_kv_heads * head_dim * 4.0, (attn_fwd_kernel<128, 4, true>),
               dim3(p.n_items, n_kv_heads), dim3(256), 0, st, p);
    FO1_LAUNCH("attn_decode_combine", (double)n_q_heads * head_dim * 8.0, attn_decode_combine_kernel<128>, dim3(n_q_heads), dim3(128), 0, st,
               (const float*)workspace, (const int*)dyn_kv_len, 64, n_kv_heads, group, (uint16_t*)out, (const int*)nullptr, 0LL, 0LL);
    return FO1_OK;
}

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: test / bench build only
// A/B hook: 0 (default) = 64-key split-KV partials + combine kernel; 1 = one workgroup per (KV head, sequence) with the waves'
// partials merged in LDS (one launch up to 2048 keys; measured slower, see attn_decode_wg_kernel).
int fo1_attention_decode_set_impl(int impl) {
    if (impl < 0 || impl > 2) return fo1::set_err(FO1_ERR_ARG, "attention_decode_set_impl: %d", impl);
    fo1::g_attn_decode_impl = impl;
    return FO1_OK;
}
#endif   // FO1_ENABLE_AB

// Batched decode attention: B sequences, one new token each (q rows [B, n_q_heads*head_dim]); sequence b attends the cache rows
// [state[b][2], state[b][0]] (its slot start .. the row just written).  grid = (max chunks per slot, KV heads, B).
// Keys per split of the batched decode attention — a function of the batch size only (a sequence's partial sums must not depend on
// who shares the launch): up to 32 sequences (BatchDecoder) 64 keys per workgroup, so that even one sequence spreads over 2 x 11
// CUs; a decode pool (64 / 128 slots) has sequences to spare and gives every workgroup up to 8 tiles of 64 keys (measured at 128 slots x
// 651-715 keys, profiles/r04_pool_step_attention_chunk_ab.json: split kernel 1.11 / 0.91 / 0.90 / 0.84 ms per step at 64 / 128 / 256 / 512) — the tile loop's register
// prefetch then overlaps the next tile's loads with the MFMAs of the current one (one-tile workgroups are a load -> compute chain).
namespace fo1 {
FO1_AB_VAR g_attn_pool_chunk = 1024;     // A/B: fo1_attention_decode_set_pool_chunk.  1024 keys: a pool slot's whole context (slot_rows <= 1024) is ONE chunk —
                                         // the split kernel writes the rows itself, no combine launch (512: 23.7 + 4.3 us per layer, 1024: 22.2 + 0; profiles/r04_pool_step_attention_one_chunk.json)
FO1_AB_VAR g_attn_tiles_per_item = 2;    // 64-key tiles per item at 17..32 sequences (A/B: fo1_attention_decode_set_small_chunk(1 .. 8); profiles/r06_decode_attention_grid_order_ab.json: 1 / 2 / 3 / 4 tiles 2.96 / 2.89 / 2.91 / 2.98 ms per step at 25 sequences)
FO1_AB_VAR g_attn_small_chunk = 64;      // A/B: fo1_attention_decode_set_small_chunk (<= 32 sequences)
static inline int decode_batch_chunk(int batch) { return batch > 32 ? g_attn_pool_chunk : g_attn_small_chunk; }
}  // namespace fo1
#ifdef FO1_ENABLE_AB
int fo1_attention_decode_set_small_chunk(int keys) {
    if (keys >= 1 && keys <= 8) {          // 1 .. 8: tiles per item at 17..32 sequences (the partials do not depend on it)
        fo1::g_attn_tiles_per_item = keys;
        return FO1_OK;
    }
    if (keys < 64 || keys > 4096 || keys % 64) return fo1::set_err(FO1_ERR_ARG, "attention_decode_set_small_chunk: %d", keys);
    fo1::g_attn_small_chunk = keys;
    return FO1_OK;
}
int fo1_attention_decode_set_pool_chunk(int keys) {
    if (keys < 64 || keys > 4096 || keys % 64) return fo1::set_err(FO1_ERR_ARG, "attention_decode_set_pool_chunk: %d", keys);
    fo1::g_attn_pool_chunk = keys;
    return FO1_OK;
}
#endif
size_t fo1_attention_decode_batch_workspace_bytes(int max_kv_len, int n_kv_heads, int head_dim, int batch) {
    return (size_t)batch * fo1::cdiv(max_kv_len, fo1::decode_batch_chunk(batch)) * n_kv_heads * 16 * (head_dim + 2) * sizeof(float);
}

static int attention_decode_batch_impl(const void* q, long long q_seq_stride, const void* kcache, long long k_tok_stride, long long k_head_stride,
                                       const void* vtcache, long long vt_row_stride, void* out, long long out_seq_stride, const int32_t* state,
                                       int batch, int max_kv_len, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* workspace,
                                       size_t workspace_bytes, void* stream, bool combine) {
    using namespace fo1;
    FO1_CHECK_ARG(q && kcache && vtcache && (out || !combine) && state && workspace && batch >= 1, "attention_decode_batch: NULL operand");
    FO1_CHECK_ARG(head_dim == 128, "attention_decode_batch: head_dim %d not built (128)", head_dim);
    FO1_CHECK_ARG(n_q_heads % n_kv_heads == 0 && n_q_heads / n_kv_heads <= 16, "attention_decode_batch: at most 16 query heads per KV head");
    FO1_CHECK_ARG(vt_row_stride % 4 == 0 && k_tok_stride % 8 == 0 && k_head_stride % 8 == 0 && q_seq_stride % 8 == 0, "attention_decode_batch: bad strides");
    if (workspace_bytes < fo1_attention_decode_batch_workspace_bytes(max_kv_len, n_kv_heads, head_dim, batch))
        return set_err(FO1_ERR_WORKSPACE, "attention_decode_batch: workspace too small");
    const int group = n_q_heads / n_kv_heads;
#ifdef FO1_ENABLE_AB
    if (g_attn_decode_impl == 1) {
        AttnDecParams d;
        d.Q = (const uint16_t*)q; d.q_seq_stride = q_seq_stride;
        d.K = (const uint16_t*)kcache; d.k_tok = k_tok_stride; d.k_head = k_head_stride;
        d.VT = (const uint16_t*)vtcache; d.vt_row = vt_row_stride;
        d.O = (uint16_t*)out; d.o_seq_stride = out_seq_stride;
        d.part = (float*)workspace; d.seq_state = (const int*)state; d.dyn_kv_len = nullptr;
        d.n_kv_heads = n_kv_heads; d.group = group; d.scale = scale;
        return launch_attn_decode_wg(d, max_kv_len, batch, n_q_heads, (hipStream_t)stream);
    }
    if (g_attn_decode_impl == 2) {
        AttnDecParams d;
        d.Q = (const uint16_t*)q; d.q_seq_stride = q_seq_stride;
        d.K = (const uint16_t*)kcache; d.k_tok = k_tok_stride; d.k_head = k_head_stride;
        d.VT = (const uint16_t*)vtcache; d.vt_row = vt_row_stride;
        d.O = (uint16_t*)out; d.o_seq_stride = out_seq_stride;
        d.part = nullptr; d.part_seq_stride = 0; d.seq_state = (const int*)state; d.dyn_kv_len = nullptr;
        d.n_kv_heads = n_kv_heads; d.group = group; d.scale = scale; d.split_keys = 0; d.n_splits = 1;
        return launch_attn_decode_ws(d, max_kv_len, batch, (hipStream_t)stream);
    }
#endif
    AttnParams p;
    p.Q = (const uint16_t*)q; p.q_tok = head_dim; p.q_head = (long long)group * head_dim;
    p.K = (const uint16_t*)kcache; p.k_tok = k_tok_stride; p.k_head = k_head_stride;
    p.VT = (const uint16_t*)vtcache; p.vt_row = vt_row_stride;
    p.O = nullptr; p.o_tok = 0; p.o_head = 0;
    p.items = nullptr; p.items2 = nullptr;
    const int chunk = decode_batch_chunk(batch);
    // 17..32 sequences: an item walks 2 tiles of 64 keys and writes each tile's partial separately (see AttnParams.part_tiles) — half the
    // workgroups, the next tile's loads under the current tile's arithmetic, the SAME partials as one-tile items (the batch invariance holds)
    const int n_chunks = cdiv(max_kv_len, chunk);
    const int ptiles = (batch > 16 && batch <= 32 && chunk == 64 && combine) ? g_attn_tiles_per_item : 1;
    p.n_items = cdiv(n_chunks, ptiles); p.Hq = n_kv_heads; p.group = 1;
    p.scale = scale; p.causal = 0; p.q_row_base = nullptr;
    p.part = (float*)workspace; p.dyn_kv_len = nullptr; p.kv_chunk = chunk * ptiles;
    p.q_range_end = group; p.part_tiles = ptiles; p.grid_batch = batch;
    p.seq_state = (const int*)state; p.q_seq_stride = q_seq_stride;
    p.part_seq_stride = (long long)n_chunks * n_kv_heads * 16 * (head_dim + 2);
    p.bias = nullptr; p.wlen = 0; p.sw_ws = p.sw_shift = p.sw_nwy = p.sw_nwx = 0;
    hipStream_t st = (hipStream_t)stream;
    if (n_chunks == 1 && combine) {       // one chunk per (sequence, KV head) — the pool's slots, or a short context at any batch size -> rows written by the split kernel itself, no combine launch
        p.O = (uint16_t*)out; p.o_tok = out_seq_stride; p.o_head = head_dim;
        FO1_LAUNCH("attn_decode_one_chunk", (double)batch * max_kv_len * n_kv_heads * head_dim * 4.0, (attn_fwd_kernel<128, 4, true>),
                   dim3(n_kv_heads * batch), dim3(256), 0, st, p);
        return FO1_OK;
    }
    FO1_LAUNCH("attn_decode_split", (double)batch * max_kv_len * n_kv_heads * head_dim * 4.0, (attn_fwd_kernel<128, 4, true>),
               dim3(p.n_items * n_kv_heads * batch), dim3(256), 0, st, p);
    if (!combine) return FO1_OK;       // the consumer sums the partials itself (fo1_gemv_attn_combine_bf16)
    FO1_LAUNCH("attn_decode_combine", (double)batch * n_q_heads * head_dim * 8.0, attn_decode_combine_kernel<128>, dim3(n_q_heads, batch), dim3(128), 0,
               st, (const float*)workspace, (const int*)nullptr, chunk, n_kv_heads, group, (uint16_t*)out, (const int*)state, p.part_seq_stride,
               out_seq_stride);
    return FO1_OK;
}

int fo1_attention_decode_batch_bf16(const void* q, long long q_seq_stride, const void* kcache, long long k_tok_stride, long long k_head_stride,
                                    const void* vtcache, long long vt_row_stride, void* out, long long out_seq_stride, const int32_t* state,
                                    int batch, int max_kv_len, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return attention_decode_batch_impl(q, q_seq_stride, kcache, k_tok_stride, k_head_stride, vtcache, vt_row_stride, out, out_seq_stride, state, batch,
                                       max_kv_len, n_q_heads, n_kv_heads, head_dim, scale, workspace, workspace_bytes, stream, true);
}

// The split-KV half of fo1_attention_decode_batch_bf16 alone: per sequence, KV head and chunk of `*kv_chunk_out` keys the unnormalised fp32 rows +
// (m, l) go to `workspace` as [batch][chunks][n_kv_heads][16][head_dim + 2]; fo1_gemv_attn_combine_bf16 (the o-projection of a decode step at
// <= 2 sequences) sums them in its prologue, so the combine launch and the [batch, heads x head_dim] activation never exist.
int fo1_attention_decode_batch_partials_bf16(const void* q, long long q_seq_stride, const void* kcache, long long k_tok_stride, long long k_head_stride,
                                             const void* vtcache, long long vt_row_stride, const int32_t* state, int batch, int max_kv_len,
                                             int n_q_heads, int n_kv_heads, int head_dim, float scale, void* workspace, size_t workspace_bytes,
                                             int* kv_chunk_out, long long* part_seq_stride_out, void* stream) {
    FO1_CHECK_ARG(kv_chunk_out && part_seq_stride_out, "attention_decode_batch_partials: NULL output");
    const int chunk = fo1::decode_batch_chunk(batch);
    *kv_chunk_out = chunk;
    *part_seq_stride_out = (long long)fo1::cdiv(max_kv_len, chunk) * n_kv_heads * 16 * (head_dim + 2);
    return attention_decode_batch_impl(q, q_seq_stride, kcache, k_tok_stride, k_head_stride, vtcache, vt_row_stride, nullptr, 0, state, batch, max_kv_len,
                                       n_q_heads, n_kv_heads, head_dim, scale, workspace, workspace_bytes, stream, false);
}

}  // extern "C"
