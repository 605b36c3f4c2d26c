// window_attention.hip — DaViT's WindowAttention core on the matrix cores (round 6).  Its own translation unit: built with
// -mllvm -amdgpu-mfma-vgpr-form=1 (vlm_fo1_amd/build.py): the kernel reads every MFMA result with VALU instructions right away (softmax, the
// half-wave exchange), and with the accumulators in AGPRs hipcc copies each of them through v_accvgpr_read / write (1 900 of the kernel's
// 6 300 instructions).
#include "common.h"

namespace fo1 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- window attention, head dim 32 (round 6; DaViT's 12 x 12 windows: modeling_davit.py:225-282) ------------------------------------------------
// One WAVE per (window, head), four adjacent heads per workgroup: a window's q / k / v live in the [tokens, 3C] rows the q/k/v GEMM wrote
// (no V^T copy, no item list), 64 bytes per row and head — the four waves of a workgroup consume 256 contiguous bytes of every row.  A window
// has at most NT * 32 tokens: its k and V fragments stay in registers for all query tiles.  S^T = K Q^T in the swapped form of attn_fwd32 (a
// lane owns one query and 16 keys of a 32-key tile), online softmax over the key tiles, and the probabilities feed the PV products from the
// score registers directly (k-slot j of step t = score register 8 t + j).  V is the operand whose reduction index (keys) is the memory-
// major one: its fragments are read column-wise (2-byte LDS reads) out of the wave's staged copy of its V rows, once per window (the
// fragments stay in registers for all query tiles).  Same arithmetic as attn_fwd32: raw scores x scale x log2(e) in fp32, P = bf16(exp2(.)),
// row sums of the unrounded exponentials, O / l rounded to bf16.  HBM traffic = q, k, v once + the output; no workgroup barrier.
//
// WS > 0 (round 6, second form): the q/k/v rows are NOT window-partitioned — they are the image's pixels in raster order (the GEMM ran on the
// LayerNorm output itself), and token (iy, ix) of window (wy, wx) is pixel (wy WS + iy, wx WS + ix) of image blockIdx.z.  Tokens outside the image
// are the reference's zero padding AFTER the norm (modeling_davit.py:248-251): their q/k/v row is the projection of a zero row = the bias, read
// from `pad_row` (the layer's bf16 q/k/v bias itself).  Outputs go to the pixels' rows; padded tokens have none.  window_partition before the
// GEMM, the padded rows inside it (44 % more rows at 40 x 30), and window_reverse (+ residual: now the proj GEMM's own epilogue) are gone.
template <int WS>
__device__ __forceinline__ long long win_tok_orow(int t, long long row0, int H, int W, int wy, int wx) {
    if constexpr (WS > 0) {
        const int iy = t / WS, ix = t - iy * WS;
        const int h = wy * WS + iy, w = wx * WS + ix;
        return h < H && w < W ? row0 + (long long)h * W + w : -1ll;
    } else {
        return row0 + t;
    }
}
template <int WS>
__device__ __forceinline__ const uint16_t* win_tok_row(int t, long long row0, int H, int W, int wy, int wx, const uint16_t* qkv, long long ld, const uint16_t* pad_row) {
    const long long r = win_tok_orow<WS>(t, row0, H, W, wy, wx);
    if constexpr (WS > 0) return r >= 0 ? qkv + r * ld : pad_row;
    else return qkv + r * ld;
}
struct WinGeom { int row0, H, W, wrow0, nWy, nWx, a0, a1; };      // = fo1_img_seg of the window operators: first pixel row, H x W, (first window row), windows down / across
template <int NT, int WS>
__global__ __launch_bounds__(256) void win_attn32_kernel(const uint16_t* __restrict__ qkv, long long ld, int C, int heads, int wtok,
                                                         uint16_t* __restrict__ out, long long ldo, uint32_t out_bytes, float c1,
                                                         int gH, int gW, const uint16_t* __restrict__ pad_row, const WinGeom* __restrict__ geoms) {
    constexpr int LDV = 80;                                   // bytes per staged V row: the fragment's two key groups in different bank halves
    __shared__ __attribute__((aligned(16))) char smem[4 * NT * 32 * LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, kg = lane >> 5;
    const int head = blockIdx.x * 4 + wave;
    if (head >= heads) return;                                // (no workgroup barrier below)
    // geometry of this window (WS > 0): image = blockIdx.z, window = blockIdx.y of the image's nWy x nWx
    int H = 0, W = 0, wy = 0, wx = 0;
    long long row0;
    if constexpr (WS > 0) {
        int nWx, nWy, irow0;
        if (geoms) { const WinGeom g = geoms[blockIdx.z]; H = g.H; W = g.W; nWy = g.nWy; nWx = g.nWx; irow0 = g.row0; }
        else { H = gH; W = gW; nWy = (H + WS - 1) / WS; nWx = (W + WS - 1) / WS; irow0 = blockIdx.z * H * W; }
        if ((int)blockIdx.y >= nWy * nWx) return;             // ragged: grid.y covers the image with the most windows
        wy = blockIdx.y / nWx; wx = blockIdx.y - wy * nWx;
        row0 = irow0;
    } else {
        row0 = (long long)blockIdx.y * wtok;
    }
    // global row of token t of this window, -1 for a padded token; its q/k/v row (a padded token: the bias row)
#define FO1_TOK_OROW(t) win_tok_orow<WS>((t), row0, H, W, wy, wx)
#define FO1_TOK_ROW(t) win_tok_row<WS>((t), row0, H, W, wy, wx, qkv, ld, pad_row)
    const int hoff = head * 32 + kg * 8;
    typedef __attribute__((ext_vector_type(4))) unsigned int wv4u;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);      // <= 2 GiB (host-checked)
    // the window's k fragments (lane = key, 8 channels), its V rows (for the staged copy) and the first query tile go out at once; the next
    // query tile is requested while the current one is worked on (all query tiles resident would cost one wave per SIMD of occupancy)
    uint4 kf0[NT], kf1[NT], vr[NT * 2], qf[2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint16_t* rp = FO1_TOK_ROW(min(32 * t + fi, wtok - 1)) + hoff;      // tokens past the window: a valid row, masked below
        kf0[t] = *reinterpret_cast<const uint4*>(rp + C);
        kf1[t] = *reinterpret_cast<const uint4*>(rp + C + 16);
    }
#pragma unroll
    for (int p = 0; p < NT * 2; ++p)
        vr[p] = *reinterpret_cast<const uint4*>(FO1_TOK_ROW(min(16 * p + (lane >> 2), wtok - 1)) + 2 * C + head * 32 + (lane & 3) * 8);
    {
        const uint16_t* rp = FO1_TOK_ROW(min(fi, wtok - 1)) + hoff;
#pragma unroll
        for (int c = 0; c < 2; ++c) qf[0][c] = *reinterpret_cast<const uint4*>(rp + 16 * c);
    }
    char* sv = smem + wave * (NT * 32 * LDV);
#pragma unroll
    for (int p = 0; p < NT * 2; ++p) *reinterpret_cast<uint4*>(sv + (16 * p + (lane >> 2)) * LDV + (lane & 3) * 16) = vr[p];
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();                          // wave-private tile, in-order LDS: only the compiler is held to the order
    __builtin_amdgcn_sched_barrier(0);
    // V fragments: step st = 16 keys (tile st / 2, half st % 2); k-slot j <-> the key of score register 8 (st % 2) + j
    bf16x8 vf[NT * 2];
#pragma unroll
    for (int st = 0; st < NT * 2; ++st) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r0 = 8 * (st & 1) + 2 * u, r1 = r0 + 1;
            const int key0 = 32 * (st >> 1) + (r0 & 3) + 8 * (r0 >> 2) + 4 * kg, key1 = 32 * (st >> 1) + (r1 & 3) + 8 * (r1 >> 2) + 4 * kg;
            w[u] = (uint32_t)*reinterpret_cast<const uint16_t*>(sv + key0 * LDV + 2 * fi) |
                   ((uint32_t)*reinterpret_cast<const uint16_t*>(sv + key1 * LDV + 2 * fi) << 16);
        }
        const uint4 pk = uint4{w[0], w[1], w[2], w[3]};
        vf[st] = __builtin_bit_cast(bf16x8, pk);
    }
#pragma unroll
    for (int qi = 0; qi < NT; ++qi) {
        if (qi + 1 < NT) {                                    // next query tile (unconditional loads: past the window they re-read its last row)
            const uint16_t* rp = FO1_TOK_ROW(min(32 * (qi + 1) + fi, wtok - 1)) + hoff;
#pragma unroll
            for (int c = 0; c < 2; ++c) qf[(qi + 1) & 1][c] = *reinterpret_cast<const uint4*>(rp + 16 * c);
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // (no skip of query tiles past the window: a branch around the stores below costs more than the tile — see the store comment)
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (32 * t < wtok) {                          // wave-uniform: key tiles past the window do not exist
                    f32x16 sc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf0[t]), __builtin_bit_cast(bf16x8, qf[qi & 1][0]), sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf1[t]), __builtin_bit_cast(bf16x8, qf[qi & 1][1]), sc, 0, 0, 0);
                    // sc[r] = raw score(key 32 t + (r & 3) + 8 (r >> 2) + 4 kg, query 32 qi + fi)
                    if (32 * t + 32 > wtok) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[r] = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg < wtok ? sc[r] : -INFINITY;
                    }
                    float mx = sc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const float m_new = fmaxf(m_run, mx * c1);       // finite: the tile's first key is valid
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    if (t > 0 && __any(alpha != 1.0f)) {          // wave-uniform: later tiles rarely move the maximum
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[r] *= alpha;
                    }
                    float ps = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[h * 8 + 2 * j], c1, -m_new));       // masked: exp2(-inf) = 0
                            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[h * 8 + 2 * j + 1], c1, -m_new));
                            ps += e0 + e1;
                            w[j] = pack_bf16x2(e0, e1);
                        }
                        const uint4 pk = uint4{w[0], w[1], w[2], w[3]};
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * t + h], __builtin_bit_cast(bf16x8, pk), o, 0, 0, 0);
                    }
                    l_run = l_run * alpha + ps;
                }
            }
            l_run += __shfl_xor(l_run, 32, 64);
            const float inv = 1.0f / l_run;
            // o[r] = channel (r / 4) * 8 + kg * 4 + r % 4 of query 32 qi + fi: the lower half-wave keeps channels 0..15, the upper 16..31
            // (bit selects, not `kg ? o[u] : o[8 + u]`: on the vector type that becomes a run-time element index, which hipcc expands into
            // a 16-step compare / select chain per element)
            const uint32_t km = kg ? 0xffffffffu : 0u;
            auto sel = [km](float a, float b) __attribute__((always_inline)) {       // kg ? a : b
                return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, a) & km) | (__builtin_bit_cast(uint32_t, b) & ~km));
            };
            float rcv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rcv[u] = __shfl_xor(sel(o[u], o[8 + u]), 32, 64);
            float v[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = sel(rcv[u], o[u]) * inv;
                v[4 + u] = sel(o[8 + u], rcv[u]) * inv;
                v[8 + u] = sel(rcv[4 + u], o[4 + u]) * inv;
                v[12 + u] = sel(o[12 + u], rcv[4 + u]) * inv;
            }
            // stores through a buffer descriptor: a lane past the window stores outside the descriptor and the hardware drops it — a branch
            // around the stores would make hipcc's wait for the next query tile (requested before them) wait for the stores themselves
            const int qrow = 32 * qi + fi;
            const long long orow = FO1_TOK_OROW(min(qrow, wtok - 1));
            const uint32_t off = qrow < wtok && orow >= 0 ? (uint32_t)((orow * ldo + head * 32 + kg * 16) * 2) : 0xC0000000u;
            __builtin_amdgcn_raw_buffer_store_b128(wv4u{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                                                   rs_o, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(wv4u{pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])},
                                                   rs_o, off + 16u, 0, 0);
        }
    }
}

#undef FO1_TOK_OROW
#undef FO1_TOK_ROW

}  // namespace fo1

extern "C" {

int fo1_window_attention_bf16(const void* qkv, long long ld, int C, int n_heads, int window_tokens, int n_windows, void* out, long long ldo,
                              float scale, void* stream) {
    using namespace fo1;
    if (n_windows == 0) return FO1_OK;
    FO1_CHECK_ARG(qkv && out, "window_attention: NULL operand");
    FO1_CHECK_ARG(n_heads > 0 && C == n_heads * 32, "window_attention: built for head dim 32 (C=%d, %d heads)", C, n_heads);
    FO1_CHECK_ARG(window_tokens >= 1 && window_tokens <= 160, "window_attention: %d tokens per window (1..160)", window_tokens);
    FO1_CHECK_ARG(n_windows > 0 && n_windows <= 65535, "window_attention: %d windows (grid.y: at most 65535)", n_windows);
    FO1_CHECK_ARG(ld >= 3 * C && ld % 8 == 0 && ldo >= C && ldo % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0,
                  "window_attention: rows must be 16-byte aligned (ld / ldo %% 8, pointers)");
    const long long out_bytes = ((long long)n_windows * window_tokens - 1) * ldo * 2 + (long long)C * 2;
    FO1_CHECK_ARG(out_bytes <= (1ll << 31), "window_attention: the output spans %lld bytes (32-bit store offsets: at most 2 GiB)", out_bytes);
    const float c1 = scale * 1.4426950408889634f;
    const double flops = 4.0 * C * (double)n_windows * window_tokens * window_tokens;
    FO1_LAUNCH("win_attn32", flops, (win_attn32_kernel<5, 0>), dim3(cdiv(n_heads, 4), n_windows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, ld, C,
               n_heads, window_tokens, (uint16_t*)out, ldo, (uint32_t)out_bytes, c1, 0, 0, (const uint16_t*)nullptr, (const WinGeom*)nullptr);
    return FO1_OK;
}

static int window_attention_map(const void* qkv, long long ld, int C, int n_heads, int window, int H, int W, int n_img, const void* geoms, int max_windows,
                                long long total_rows, const void* pad_row, void* out, long long ldo, float scale, void* stream, const char* what) {
    using namespace fo1;
    if (n_img == 0 || total_rows == 0) return FO1_OK;
    FO1_CHECK_ARG(qkv && out && pad_row, "%s: NULL operand", what);
    FO1_CHECK_ARG(n_heads > 0 && C == n_heads * 32, "%s: built for head dim 32 (C=%d, %d heads)", what, C, n_heads);
    FO1_CHECK_ARG(window == 12, "%s: built for 12 x 12 windows (got %d)", what, window);
    FO1_CHECK_ARG(n_img >= 1 && n_img <= 65535 && max_windows >= 1 && max_windows <= 65535, "%s: %d images, %d windows per image (grid: at most 65535 each)", what, n_img, max_windows);
    FO1_CHECK_ARG(ld >= 3 * C && ld % 8 == 0 && ldo >= C && ldo % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)pad_row & 15) == 0,
                  "%s: rows must be 16-byte aligned (ld / ldo %% 8, pointers)", what);
    const long long out_bytes = (total_rows - 1) * ldo * 2 + (long long)C * 2;
    FO1_CHECK_ARG(out_bytes <= (1ll << 31) && total_rows < (1ll << 31), "%s: the output spans %lld bytes (32-bit store offsets: at most 2 GiB)", what, out_bytes);
    const double flops = 4.0 * C * (double)n_img * max_windows * 144.0 * 144.0;
    FO1_LAUNCH("win_attn32", flops, (win_attn32_kernel<5, 12>), dim3(cdiv(n_heads, 4), max_windows, n_img), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, ld,
               C, n_heads, 144, (uint16_t*)out, ldo, (uint32_t)out_bytes, scale * 1.4426950408889634f, H, W, (const uint16_t*)pad_row, (const WinGeom*)geoms);
    return FO1_OK;
}

int fo1_window_attention_map_bf16(const void* qkv, long long ld, int C, int n_heads, int window, int H, int W, int batch, const void* pad_row, void* out,
                                  long long ldo, float scale, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(H > 0 && W > 0 && batch >= 1, "window_attention_map: bad shape %dx%d x%d", H, W, batch);
    return window_attention_map(qkv, ld, C, n_heads, window, H, W, batch, nullptr, cdiv(H, 12) * cdiv(W, 12), (long long)batch * H * W, pad_row, out, ldo, scale,
                                stream, "window_attention_map");
}

int fo1_window_attention_map_var_bf16(const void* qkv, long long ld, int C, int n_heads, int window, const void* segs, int n_img, int max_windows,
                                      long long total_pixels, const void* pad_row, void* out, long long ldo, float scale, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(segs != nullptr, "window_attention_map_var: need the device fo1_img_seg table of the window geometry");
    return window_attention_map(qkv, ld, C, n_heads, window, 0, 0, n_img, segs, max_windows, total_pixels, pad_row, out, ldo, scale, stream, "window_attention_map_var");
}

}  // extern "C"
