// hfre.hip — Hybrid Fine-grained Region Encoder pooling for gfx950 (MI355X).
//
// Replaces HFREModule.__call__ / extract_vt_region_feature / gen_sineembed_for_position
// (reference hybrid_finegrained_region_encoder.py:55-103,230-273,275-469) and the
// torchvision roi_align + F.interpolate + torch.cat chain inside them.
//
// HBM-bound gather, no MFMA: every (box, source, channel-chunk, row-slice) is one
// 256-thread workgroup that
//   1. rebuilds the per-axis tap weights of the box in LDS (hfre_math.h),
//   2. streams the footprint pixels of the bf16 token-major map, 16 B per lane,
//      channels across lanes (coalesced: consecutive lanes = consecutive channels,
//      consecutive pixel slots = consecutive pixels), fp32 accumulate,
//   3. reduces pixel slots with wave shuffles and waves through LDS,
//   4. writes one fp32 partial row to the workspace.
// hfre_finish_kernel sums the row-slices of each box in a fixed order, adds the
// sine box embedding and writes the fp32 output.  No atomics: results are
// run-to-run deterministic.
#include "common.h"
#include "ab.h"
#include "hfre_math.h"

namespace fo1 {

constexpr int kHfreThreads = 256;
constexpr int kHfreWaves = kHfreThreads / 64;
constexpr int kHfreWThreads = 64;    // hfre_weights_kernel: one wave per (box, source) — a latency chain (box load, atomic, two phases), so
                                     // what matters is how many are resident: 6 000 workgroups at 12 images x 100 boxes
constexpr int kHfreMaxChunk = 512;   // channels per workgroup (64 lanes x 8 bf16)
[[maybe_unused]] constexpr int kHfreUnroll = 8;

struct HfreSrcDev {
    const uint16_t* data;
    int H, W, C, ld, roi_H, roi_W;
    float scale;
    int box_space, out_offset;
    int chunk, nchunks, max_slices;
    int wg_base;  // first workgroup id of this source
    int ws_off;   // float offset of this source inside one box's workspace block
    // ---- band path (hfre_pool_bands_kernel) ----
    int lpp;          // lanes per pixel: a workgroup streams lpp * 8 channels of the map (16 B per lane)
    int rb;           // rows staged in LDS at a time (one band)
    int rr;           // rows per work item (a row range = one partial-sum slice of every box that touches it)
    int n_ranges;     // ceil(H / rr)
    int bchunks;      // C / (lpp * 8)
    int item_base;    // first work item of this source; items of one image = bchunks * n_ranges
};

struct HfreParams {
    HfreSrcDev src[FO1_HFRE_MAX_SOURCES];
    int n_sources;
    const float* boxes;     // aux space
    const float* boxes_vt;  // vt space or nullptr (then aux * (vsx, vsy))
    int n_boxes;
    float vsx, vsy;
    int P;
    int pos_mode;
    float pos_w, pos_h;
    float* out;
    int out_ld, region_dim;
    float* ws;
    int ws_box_stride;  // floats per box
    int pixel_budget;
    float* wbuf;        // per (box, source): wy[kHfreWStride] | wx[kHfreWStride] tap weights on the source map
    int* hdr;           // per (box, source): r_lo, r_hi, c_lo, c_hi, rows_per_slice, n_slices, -, -
    // ---- work-list path (fo1_hfre_region_pool_ex) ----
    const int* box_image;                       // image of each box (batched call) or nullptr
    long long img_stride[FO1_HFRE_MAX_SOURCES]; // elements between consecutive images of a source map
    int* n_items;                               // kHfreBuckets device counters of work items, kHfreCtrStride ints apart (zero on entry; the
                                                // finish kernel re-zeroes them).  ONE counter serialises: 6 000 returning atomics on one
                                                // address cost 123 us at 12 images x 100 boxes (profiles/r02_bench_default.json, hfre_weights)
    int2* items;                                // {box, source << 24 | chunk << 12 | slice}: only the slices that exist; bucket b owns
                                                // [b * items_cap, (b + 1) * items_cap)
    int items_cap;                              // capacity of ONE bucket (worst case); the walk never reads past it
    int ln_on, ln_split;                        // region LayerNorm (reference :365-372): blocks [0, ln_split) and [ln_split, region_dim)
    const float* ln_w0; const float* ln_b0; const float* ln_w1; const float* ln_b1;
    float ln_eps;
    uint16_t* out_bf16; int out_bf16_ld;        // optional second destination: the same rows cast to bf16 (RNE) — what encode_regions does
                                                // before mm_projector_aux (omchat_qwen2_5_vl.py:106): the cast rides in the finish kernel
    int band_mode;                              // 1: hfre_pool_bands_kernel (slices = row ranges of rr rows), 0: work list (slices by pixel budget)
    int n_images, band_items;                   // band path: images in the call, total work items
    int bucket_per;                             // work-list order: > 0 = (box, source) pair q appends to bucket q / bucket_per (the list is then walked
                                                // box-major = IMAGE-major: what is in flight at any time reads ONE image's maps, 55 MB of the 256 MB
                                                // Infinity Cache, so the overlapping proposals' re-reads stop at the cache instead of HBM); 0 = q % 64 (rounds 2-5)
    float* dimt;                                // work-list path: dim_t table of the sine embedding, region_dim / 8 entries (written by
                                                // hfre_weights_kernel, read by hfre_finish_vec_kernel); nullptr = powf per element
};

constexpr int kHfreWStride = FO1_HFRE_MAX_EXTENT;
constexpr int kHfreBuckets = 64;      // work-list buckets: (box, source) pair q appends to bucket q % 64
constexpr int kHfreCtrStride = 64;    // ints between bucket counters (256 B: one counter per memory channel line)

struct Footprint {
    RoiAxis ay, ax;
    int r_lo, r_hi, c_lo, c_hi;  // inclusive, on the source map
    int rows_per_slice, n_slices;
};

__device__ __forceinline__ void load_box(const HfreParams& p, int n, bool vt, float& x1, float& y1, float& x2, float& y2) {
    const bool direct = vt && p.boxes_vt != nullptr;
    const float4 b = reinterpret_cast<const float4*>(direct ? p.boxes_vt : p.boxes)[n];
    x1 = b.x; y1 = b.y; x2 = b.z; y2 = b.w;
    if (vt && !direct) {
        x1 *= p.vsx; x2 *= p.vsx;
        y1 *= p.vsy; y2 *= p.vsy;
    }
}

__device__ __forceinline__ Footprint hfre_footprint(const HfreParams& p, const HfreSrcDev& s, int n) {
    float x1, y1, x2, y2;
    load_box(p, n, s.box_space == 1, x1, y1, x2, y2);
    Footprint f;
    f.ay = make_roi_axis(y1, y2, s.scale, p.P, s.roi_H);
    f.ax = make_roi_axis(x1, x2, s.scale, p.P, s.roi_W);
    upsample_range(f.ay.lo, f.ay.hi, s.H, s.roi_H, f.r_lo, f.r_hi);
    upsample_range(f.ax.lo, f.ax.hi, s.W, s.roi_W, f.c_lo, f.c_hi);
    const int fh = f.r_hi - f.r_lo + 1, fw = f.c_hi - f.c_lo + 1;
    if (fh <= 0 || fw <= 0) {
        f.rows_per_slice = 1;
        f.n_slices = 0;
    } else if (p.band_mode) {
        f.rows_per_slice = s.rr;                              // slice k = the box's rows inside row range (r_lo / rr + k)
        f.n_slices = f.r_hi / s.rr - f.r_lo / s.rr + 1;
    } else {
        f.rows_per_slice = slice_rows(fw, p.pixel_budget);
        f.n_slices = (fh + f.rows_per_slice - 1) / f.rows_per_slice;
    }
    return f;
}

// Stage 1: one workgroup per (box, source) builds the per-axis tap weights on the source map once
// (roi_align taps composed with the bilinear-upsample taps) and a small header; every pooling workgroup of
// that (box, source) then just loads them.
__global__ __launch_bounds__(kHfreWThreads) void hfre_weights_kernel(const HfreParams p) {
    __shared__ float s_wAy[FO1_HFRE_MAX_EXTENT];
    __shared__ float s_wAx[FO1_HFRE_MAX_EXTENT];
    const int n = blockIdx.x / p.n_sources, si = blockIdx.x - n * p.n_sources;
    const HfreSrcDev& s = p.src[si];
    const Footprint f = hfre_footprint(p, s, n);
    const int tid = threadIdx.x;
    int* h = p.hdr + (size_t)blockIdx.x * 8;
    if (tid == 0) {
        h[0] = f.r_lo; h[1] = f.r_hi; h[2] = f.c_lo; h[3] = f.c_hi; h[4] = f.rows_per_slice; h[5] = f.n_slices;
    }
    if (p.dimt != nullptr && p.pos_mode != 0) {
        // dim_t[j] = 10000^(2j / d) of gen_sineembed_for_position (reference :55-103): a function of the channel pair only
        const int d = p.region_dim / 4;
        for (int j = blockIdx.x * kHfreWThreads + tid; j < d / 2; j += gridDim.x * kHfreWThreads)
            p.dimt[j] = powf(10000.0f, (float)(2 * j) / (float)d);
    }
    if (f.n_slices == 0) return;
    if (p.items != nullptr) {
        // work list: reserve this (box, source)'s chunk x slice items in one atomic; the order of the list varies run to run, the
        // results do not (every item owns its partial row, the finish sums in slice order)
        __shared__ int s_base;
        const int cnt = f.n_slices * s.nchunks;
        const int bucket = p.bucket_per > 0 ? (int)blockIdx.x / p.bucket_per : (int)blockIdx.x % kHfreBuckets;
        if (tid == 0) s_base = atomicAdd(p.n_items + bucket * kHfreCtrStride, cnt);
        __syncthreads();
        int2* list = p.items + (size_t)bucket * p.items_cap;
        for (int j = tid; j < cnt; j += kHfreWThreads) {
            const int ch = j / f.n_slices, k = j - ch * f.n_slices;
            if (s_base + j < p.items_cap) list[s_base + j] = make_int2(n, (si << 24) | (ch << 12) | k);
        }
    }
    const bool up_y = (s.H != s.roi_H), up_x = (s.W != s.roi_W);
    if (up_y)
        for (int a = f.ay.lo + tid; a <= f.ay.hi; a += kHfreWThreads) s_wAy[a - f.ay.lo] = roi_axis_weight(f.ay, a);
    if (up_x)
        for (int a = f.ax.lo + tid; a <= f.ax.hi; a += kHfreWThreads) s_wAx[a - f.ax.lo] = roi_axis_weight(f.ax, a);
    __syncthreads();
    float* wy = p.wbuf + (size_t)blockIdx.x * 2 * kHfreWStride;
    float* wx = wy + kHfreWStride;
    const int fh = f.r_hi - f.r_lo + 1, fw = f.c_hi - f.c_lo + 1;
    for (int r = tid; r < fh; r += kHfreWThreads)
        wy[r] = up_y ? upsample_axis_weight(f.r_lo + r, s_wAy, f.ay.lo, f.ay.hi, s.H, s.roi_H) : roi_axis_weight(f.ay, f.r_lo + r);
    for (int c = tid; c < fw; c += kHfreWThreads)
        wx[c] = up_x ? upsample_axis_weight(f.c_lo + c, s_wAx, f.ax.lo, f.ax.hi, s.W, s.roi_W) : roi_axis_weight(f.ax, f.c_lo + c);
}

#ifdef FO1_ENABLE_AB      // round-1 worst-case-grid form (nine workgroups in ten only read a header and leave): A/B only
__global__ __launch_bounds__(kHfreThreads) void hfre_pool_kernel(const HfreParams p) {
    __shared__ float s_wy[FO1_HFRE_MAX_EXTENT];
    __shared__ float s_wx[FO1_HFRE_MAX_EXTENT];
    __shared__ float s_red[kHfreWaves][kHfreMaxChunk];

    const int bid = blockIdx.x;
    int si = 0;
    for (int i = 1; i < p.n_sources; ++i)
        if (bid >= p.src[i].wg_base) si = i;
    const HfreSrcDev& s = p.src[si];
    int local = bid - s.wg_base;
    const int k = local % s.max_slices;
    local /= s.max_slices;
    const int chunk_id = local % s.nchunks;
    const int n = local / s.nchunks;
    if (n >= p.n_boxes) return;

    const int* h = p.hdr + ((size_t)n * p.n_sources + si) * 8;
    const int n_slices = h[5];
    if (k >= n_slices) return;  // uniform across the workgroup
    const int r_lo = h[0], r_hi = h[1], c_lo = h[2], c_hi = h[3], rps = h[4];

    const int tid = threadIdx.x;
    const int row0 = r_lo + k * rps;
    int row1 = row0 + rps - 1;
    if (row1 > r_hi) row1 = r_hi;
    const int nrows = row1 - row0 + 1;
    const int fw = c_hi - c_lo + 1;

    // ---- per-axis weights: coalesced copy of this slice's rows / the footprint's columns ------
    const float* gwy = p.wbuf + ((size_t)n * p.n_sources + si) * 2 * kHfreWStride + (row0 - r_lo);
    const float* gwx = p.wbuf + ((size_t)n * p.n_sources + si) * 2 * kHfreWStride + kHfreWStride;
    for (int r = tid; r < nrows; r += kHfreThreads) s_wy[r] = gwy[r];
    for (int c = tid; c < fw; c += kHfreThreads) s_wx[c] = gwx[c];
    __syncthreads();

    struct { int c_lo; } f = {c_lo};

    // ---- stream the footprint ----------------------------------------------
    const int lpp = s.chunk >> 3;        // lanes per pixel (8 bf16 = 16 B per lane), power of two
    const int lane = tid & 63, wave = tid >> 6;
    const int spw = 64 / lpp;            // pixel slots per wave
    const int slot = wave * spw + lane / lpp;
    const int cl = lane & (lpp - 1);
    const int slots = kHfreWaves * spw;
    const int npix = nrows * fw;
    const uint16_t* base = s.data + (size_t)chunk_id * s.chunk + (size_t)cl * 8;

    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;

    for (int i0 = slot; i0 < npix; i0 += slots * kHfreUnroll) {
        uint4 v[kHfreUnroll];
        float w[kHfreUnroll];
#pragma unroll
        for (int u = 0; u < kHfreUnroll; ++u) {
            const int idx = i0 + u * slots;
            const bool ok = idx < npix;
            const int id2 = ok ? idx : i0;
            const int r = id2 / fw;
            const int c = id2 - r * fw;
            w[u] = ok ? s_wy[r] * s_wx[c] : 0.0f;
            const size_t pix = (size_t)(row0 + r) * s.W + (size_t)(f.c_lo + c);
            v[u] = *reinterpret_cast<const uint4*>(base + pix * s.ld);
        }
#pragma unroll
        for (int u = 0; u < kHfreUnroll; ++u) {
            acc[0] = fmaf(w[u], bf16_lo(v[u].x), acc[0]);
            acc[1] = fmaf(w[u], bf16_hi(v[u].x), acc[1]);
            acc[2] = fmaf(w[u], bf16_lo(v[u].y), acc[2]);
            acc[3] = fmaf(w[u], bf16_hi(v[u].y), acc[3]);
            acc[4] = fmaf(w[u], bf16_lo(v[u].z), acc[4]);
            acc[5] = fmaf(w[u], bf16_hi(v[u].z), acc[5]);
            acc[6] = fmaf(w[u], bf16_lo(v[u].w), acc[6]);
            acc[7] = fmaf(w[u], bf16_hi(v[u].w), acc[7]);
        }
    }

    // ---- reduce pixel slots inside the wave, then waves through LDS ----------
    for (int off = 32; off >= lpp; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    }
    if (lane < lpp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_red[wave][cl * 8 + j] = acc[j];
    }
    __syncthreads();
    float* wsrow = p.ws + (size_t)n * p.ws_box_stride + s.ws_off + (size_t)k * s.C + (size_t)chunk_id * s.chunk;
    for (int c = tid; c < s.chunk; c += kHfreThreads) {
        float t = s_red[0][c];
#pragma unroll
        for (int wv = 1; wv < kHfreWaves; ++wv) t += s_red[wv][c];
        wsrow[c] = t;
    }
}

__global__ __launch_bounds__(256) void hfre_finish_kernel(const HfreParams p) {
    __shared__ int s_nsl[FO1_HFRE_MAX_SOURCES];
    __shared__ float s_box[4];
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < p.n_sources) s_nsl[tid] = p.hdr[((size_t)n * p.n_sources + tid) * 8 + 5];
    if (tid == 32 && p.pos_mode != 0) {
        float x1, y1, x2, y2;
        load_box(p, n, p.pos_mode == 1, x1, y1, x2, y2);
        // reference :457-463 — normalise, xyxy -> cxcywh (fp32, same op order)
        x1 = x1 / p.pos_w; x2 = x2 / p.pos_w;
        y1 = y1 / p.pos_h; y2 = y2 / p.pos_h;
        const float w = x2 - x1, h = y2 - y1;
        s_box[0] = x1 + w / 2.0f;  // cx
        s_box[1] = y1 + h / 2.0f;  // cy
        s_box[2] = w;
        s_box[3] = h;
    }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + tid;
    if (c >= p.region_dim) return;
    float v = 0.0f;
    for (int i = 0; i < p.n_sources; ++i) {
        const HfreSrcDev& s = p.src[i];
        if (c >= s.out_offset && c < s.out_offset + s.C) {
            const float* w = p.ws + (size_t)n * p.ws_box_stride + s.ws_off + (c - s.out_offset);
            const int nsl = s_nsl[i];
            for (int k = 0; k < nsl; ++k) v += w[(size_t)k * s.C];
        }
    }
    if (p.pos_mode != 0) {
        // gen_sineembed_for_position (reference :55-103): blocks ordered (y, x, w, h)
        const int d = p.region_dim / 4;
        const int q = c / d, i = c - q * d;
        const float coord = (q == 0) ? s_box[1] : (q == 1) ? s_box[0] : s_box[q];
        const float dim_t = powf(10000.0f, (float)(2 * (i / 2)) / (float)d);
        const float ang = coord * 6.283185307179586f / dim_t;
        v += (i & 1) ? cosf(ang) : sinf(ang);
    }
    p.out[(size_t)n * p.out_ld + c] = v;
    if (p.out_bf16) p.out_bf16[(size_t)n * p.out_bf16_ld + c] = f32_to_bf16(v);
}

#endif   // FO1_ENABLE_AB (hfre_pool_kernel, hfre_finish_kernel)

// ------------------------------------------------------------------------------------------
// Work-list form (fo1_hfre_region_pool_ex): the grid of hfre_pool_kernel is sized for the worst case (every box as tall as the map:
// ~147 workgroups per box at 640x480), of which a typical proposal uses ~15 — nine out of ten workgroups only read their header and
// leave.  Here hfre_weights_kernel also appends the (box, source, chunk, slice) tuples that exist to a device list, and
// hfre_pool_items_kernel walks the list grid-stride with a fixed number of workgroups: no empty workgroups, all boxes of all
// images of a batch in one launch.  hfre_finish2_kernel sums the slices in slice order, applies the region LayerNorm when
// configured and adds the sine box embedding.  (A single-launch variant — per-box arrival ticket, last workgroup finishes the row —
// was measured and dropped: in-workgroup tap weights are recomputed by every chunk x slice workgroup and a release per workgroup
// costs microseconds; 107-127 us vs 59 us for this three-kernel form at 100 boxes, profiles/r02_hfre_sweep.md.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float hfre_sine(const HfreParams& p, const float* s_box, int c) {
    // gen_sineembed_for_position (reference :55-103): blocks ordered (y, x, w, h)
    const int d = p.region_dim / 4;
    const int qd = c / d, i = c - qd * d;
    const float coord = (qd == 0) ? s_box[1] : (qd == 1) ? s_box[0] : s_box[qd];
    const float dim_t = powf(10000.0f, (float)(2 * (i / 2)) / (float)d);
    const float ang = coord * 6.283185307179586f / dim_t;
    return (i & 1) ? cosf(ang) : sinf(ang);
}

// One workgroup (kHfreThreads) finishes output row n.  s_misc: >= FO1_HFRE_MAX_SOURCES ints, s_stat: 8 floats (LDS).
__device__ __forceinline__ void hfre_finish_row(const HfreParams& p, int n, int seg, int nseg, int* s_misc, float* s_stat) {
    const int tid = threadIdx.x;
    if (tid < p.n_sources) s_misc[tid] = p.hdr[((size_t)n * p.n_sources + tid) * 8 + 5];
    if (tid == 32 && p.pos_mode != 0) {
        float x1, y1, x2, y2;
        load_box(p, n, p.pos_mode == 1, x1, y1, x2, y2);
        x1 = x1 / p.pos_w; x2 = x2 / p.pos_w;       // reference :457-463 — normalise, xyxy -> cxcywh (fp32, same op order)
        y1 = y1 / p.pos_h; y2 = y2 / p.pos_h;
        const float w = x2 - x1, h = y2 - y1;
        s_stat[0] = x1 + w / 2.0f;
        s_stat[1] = y1 + h / 2.0f;
        s_stat[2] = w;
        s_stat[3] = h;
    }
    __syncthreads();
    // the row is processed in the (at most two) LayerNorm blocks; without LayerNorm it is one block
    const int nblk = (p.ln_on && p.ln_split > 0 && p.ln_split < p.region_dim) ? 2 : 1;
    for (int b = 0; b < nblk; ++b) {
        const int c0 = (b == 0) ? 0 : p.ln_split;
        const int c1 = (nblk == 2 && b == 0) ? p.ln_split : p.region_dim;
        float mean = 0.f, rstd = 1.f;
        for (int pass = 0; pass < (p.ln_on ? 3 : 1); ++pass) {
            // pass 0: slice sums [+ block sum]; pass 1: variance (two-pass, like nn.LayerNorm); pass 2: normalise + emit
            float part = 0.f;
            for (int c = c0 + seg * kHfreThreads + tid; c < c1; c += nseg * kHfreThreads) {   // nseg > 1 only without LayerNorm
                float v = 0.0f;
                for (int i = 0; i < p.n_sources; ++i) {
                    const HfreSrcDev& q = p.src[i];
                    if (c >= q.out_offset && c < q.out_offset + q.C) {
                        const float* w = p.ws + (size_t)n * p.ws_box_stride + q.ws_off + (c - q.out_offset);
                        const int nsl = s_misc[i];
                        for (int kk = 0; kk < nsl; ++kk) v += w[(size_t)kk * q.C];
                    }
                }
                if (!p.ln_on) {
                    if (p.pos_mode != 0) v += hfre_sine(p, s_stat, c);
                    p.out[(size_t)n * p.out_ld + c] = v;
                    if (p.out_bf16) p.out_bf16[(size_t)n * p.out_bf16_ld + c] = f32_to_bf16(v);
                } else if (pass == 0) {
                    part += v;
                } else if (pass == 1) {
                    part += (v - mean) * (v - mean);
                } else {
                    const bool first = (c0 == 0 && p.ln_split > 0);      // block [0, ln_split) -> (w0, b0); otherwise (w1, b1)
                    const float* lw = first ? p.ln_w0 : p.ln_w1;
                    const float* lb = first ? p.ln_b0 : p.ln_b1;
                    float o = (v - mean) * rstd * lw[c - c0] + lb[c - c0];
                    if (p.pos_mode != 0) o += hfre_sine(p, s_stat, c);
                    p.out[(size_t)n * p.out_ld + c] = o;
                    if (p.out_bf16) p.out_bf16[(size_t)n * p.out_bf16_ld + c] = f32_to_bf16(o);
                }
            }
            if (p.ln_on && pass < 2) {
                // block-wide fixed-order reduction of `part`
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                __syncthreads();
                if ((tid & 63) == 0) s_stat[4 + (tid >> 6)] = part;
                __syncthreads();
                const float tot = (s_stat[4] + s_stat[5]) + (s_stat[6] + s_stat[7]);
                if (pass == 0) mean = tot / (float)(c1 - c0);
                else rstd = rsqrtf(tot / (float)(c1 - c0) + p.ln_eps);
            }
        }
    }
}

template <int UNROLL>
__global__ __launch_bounds__(kHfreThreads) void hfre_pool_items_kernel(const HfreParams p) {
    __shared__ float s_wy[FO1_HFRE_MAX_EXTENT];
    __shared__ float s_wx[FO1_HFRE_MAX_EXTENT];
    __shared__ float s_red[kHfreWaves][kHfreMaxChunk];
    __shared__ int s_end[kHfreBuckets];            // inclusive prefix of the bucket counts
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid < kHfreBuckets) {
        int c = p.n_items[tid * kHfreCtrStride];
        if (c > p.items_cap) c = p.items_cap;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(c, o, 64);
            if (lane >= o) c += t;
        }
        s_end[tid] = c;
    }
    __syncthreads();
    const int n_items = s_end[kHfreBuckets - 1];
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        int b = 0;                                 // first bucket whose inclusive prefix exceeds `it` (uniform over the workgroup)
#pragma unroll
        for (int o = kHfreBuckets / 2; o > 0; o >>= 1)
            if (s_end[b + o - 1] <= it) b += o;
        const int2 item = p.items[(size_t)b * p.items_cap + (it - (b ? s_end[b - 1] : 0))];
        const int n = item.x, si = item.y >> 24, chunk_id = (item.y >> 12) & 0xFFF, k = item.y & 0xFFF;
        const HfreSrcDev& s = p.src[si];
        const int* h = p.hdr + ((size_t)n * p.n_sources + si) * 8;
        const int r_lo = h[0], r_hi = h[1], c_lo = h[2], c_hi = h[3], rps = h[4];
        const int img = p.box_image ? p.box_image[n] : 0;
        const int row0 = r_lo + k * rps;
        int row1 = row0 + rps - 1;
        if (row1 > r_hi) row1 = r_hi;
        const int nrows = row1 - row0 + 1;
        const int fw = c_hi - c_lo + 1;
        const float* gwy = p.wbuf + ((size_t)n * p.n_sources + si) * 2 * kHfreWStride + (row0 - r_lo);
        const float* gwx = p.wbuf + ((size_t)n * p.n_sources + si) * 2 * kHfreWStride + kHfreWStride;
        for (int r = tid; r < nrows; r += kHfreThreads) s_wy[r] = gwy[r];
        for (int c = tid; c < fw; c += kHfreThreads) s_wx[c] = gwx[c];
        __syncthreads();

        const int lpp = s.chunk >> 3;        // lanes per pixel (8 bf16 = 16 B per lane), power of two
        const int spw = 64 / lpp;            // pixel slots per wave
        const int slot = wave * spw + lane / lpp;
        const int cl = lane & (lpp - 1);
        const int slots = kHfreWaves * spw;
        const int npix = nrows * fw;
        const uint16_t* base = s.data + (long long)img * p.img_stride[si] + (size_t)chunk_id * s.chunk + (size_t)cl * 8;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        // (row, column) of the lane's pixels walk the footprint incrementally: idx advances by `slots` per load, i.e. by dq rows and
        // dr columns with one carry — one integer division per item instead of one per 16-byte load (~25 VALU instructions each,
        // in a loop whose useful work per load is 8 fmaf).  Same pixels in the same order: the sums are unchanged.
        const int dq = slots / fw, dr = slots - dq * fw;
        int pr = slot / fw, pc = slot - pr * fw;
        for (int i0 = slot; i0 < npix; i0 += slots * UNROLL) {
            uint4 v[UNROLL];
            float w[UNROLL];
            const int rf = pr, cf = pc;          // the iteration's first pixel is valid: the address of lanes past the end
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool ok = i0 + u * slots < npix;
                const int r = ok ? pr : rf;
                const int c = ok ? pc : cf;
                w[u] = ok ? s_wy[r] * s_wx[c] : 0.0f;
                const size_t pix = (size_t)(row0 + r) * s.W + (size_t)(c_lo + c);
                v[u] = *reinterpret_cast<const uint4*>(base + pix * s.ld);
                pc += dr;
                pr += dq;
                if (pc >= fw) { pc -= fw; ++pr; }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc[0] = fmaf(w[u], bf16_lo(v[u].x), acc[0]);
                acc[1] = fmaf(w[u], bf16_hi(v[u].x), acc[1]);
                acc[2] = fmaf(w[u], bf16_lo(v[u].y), acc[2]);
                acc[3] = fmaf(w[u], bf16_hi(v[u].y), acc[3]);
                acc[4] = fmaf(w[u], bf16_lo(v[u].z), acc[4]);
                acc[5] = fmaf(w[u], bf16_hi(v[u].z), acc[5]);
                acc[6] = fmaf(w[u], bf16_lo(v[u].w), acc[6]);
                acc[7] = fmaf(w[u], bf16_hi(v[u].w), acc[7]);
            }
        }
        for (int off = 32; off >= lpp; off >>= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        }
        if (lane < lpp) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_red[wave][cl * 8 + j] = acc[j];
        }
        __syncthreads();
        float* wsrow = p.ws + (size_t)n * p.ws_box_stride + s.ws_off + (size_t)k * s.C + (size_t)chunk_id * s.chunk;
        for (int c = tid; c < s.chunk; c += kHfreThreads) {
            float t = s_red[0][c];
#pragma unroll
            for (int wv = 1; wv < kHfreWaves; ++wv) t += s_red[wv][c];
            wsrow[c] = t;
        }
        __syncthreads();   // s_wy / s_wx / s_red are rewritten by the next item
    }
}

#ifdef FO1_ENABLE_AB
// ------------------------------------------------------------------------------------------
// Band form (round 4, A/B only — MEASURED 4x SLOWER than the work-list form, see below): every map row is read from HBM ONCE.
// The work-list form above gathers box by box: two proposals that overlap read the same pixels twice, and because a (box, source)'s
// items are spread over all 8 XCDs those re-reads miss the XCD L2 — PMC: 2.25 GB fetched for a 1.19 GB footprint union at 25 images x
// 100 proposals (profiles/r03_pmc_traffic.json), where the union itself is 87 % of the maps.  Here the roles are swapped: a workgroup
// owns (image, source, channel chunk of lpp * 8 channels, range of rr rows), streams that strip of the map through a double-buffered
// LDS band (rb rows at a time, the loads of the next band in flight under the arithmetic of the current one) and accumulates, for
// EVERY box of the image whose footprint touches the band, wy[r] * wx[c] * F[r, c, :] into that box's accumulator row in LDS.  At
// the end of the range each box's partial row goes to the workspace as slice k = range - first range of the box; the finish kernels
// (unchanged) sum the slices in order.  Same separable weights, same fp32 fmaf per (pixel, channel); per box the pixels are summed
// band by band, column block by column block (a re-association of the work-list form's order: results agree to fp32 rounding, and
// are run-to-run deterministic — a box's accumulator is owned by one wave).
// MEASURED (25 images x 100 proposals inside the packed pass, profiles/r04_hfre_band_form_in_pipeline_4x_slower.json): 1609 us against
// 417 us for the work-list kernel.  The strip arithmetic is cut into (box, band) units of one or two map rows (a band is what 40 KB of
// LDS holds: 184 px x 128 B = one row of the widest FPN level), ~3 x more units than the work-list form's 256-pixel slices, and every
// unit pays a dependent chain — its tap weights from L2, a 24-shuffle slot reduction, an LDS read-modify-write — with three boxes per
// wave per band in sequence: ~6 us per band where the band's bytes stream in ~1.2 us.  The work-list kernel already moves its
// (redundant) bytes at 5.4 TB/s; trading its re-reads for this bookkeeping loses.  Kept in the test / bench build for the A/B.
// ------------------------------------------------------------------------------------------
constexpr int kHfreBandThreads = 512;
constexpr int kHfreBandWaves = kHfreBandThreads / 64;
constexpr int kHfreBandBuf = 40960;                // bytes per LDS band buffer
constexpr int kHfreBandMaxBoxes = 128;             // boxes accumulated per pass (more boxes on one image: further passes over the strip)
constexpr int kHfreBandLoads = kHfreBandBuf / 16 / kHfreBandThreads;   // 16-byte loads per thread and band (5)
constexpr int kHfreBandSmem = 2 * kHfreBandBuf + kHfreBandMaxBoxes * 64 * 4;

typedef __attribute__((ext_vector_type(4))) unsigned int hf_u32x4;

__global__ __launch_bounds__(kHfreBandThreads) void hfre_pool_bands_kernel(const HfreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hb_smem[];
    unsigned char* const s_band = hb_smem;                                              // [2][kHfreBandBuf]
    float* const s_acc = reinterpret_cast<float*>(hb_smem + 2 * kHfreBandBuf);          // [kHfreBandMaxBoxes][ch]
    __shared__ int s_box[kHfreBandMaxBoxes], s_rlo[kHfreBandMaxBoxes], s_rhi[kHfreBandMaxBoxes], s_clo[kHfreBandMaxBoxes], s_chi[kHfreBandMaxBoxes];
    __shared__ int s_wcnt[kHfreBandWaves];
    __shared__ int s_total, s_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- which strip ----
    int si = 0;
    for (int i = 1; i < p.n_sources; ++i) {      // (item ranges are laid out largest strip first, not in source order)
        const int b0 = p.src[i].item_base;
        if ((int)blockIdx.x >= b0 && (int)blockIdx.x < b0 + p.n_images * p.src[i].bchunks * p.src[i].n_ranges) si = i;
    }
    const HfreSrcDev& s = p.src[si];
    int local = blockIdx.x - s.item_base;
    const int per_img = s.bchunks * s.n_ranges;
    const int img = local / per_img;
    local -= img * per_img;
    const int chunk_id = local / s.n_ranges, g = local - chunk_id * s.n_ranges;
    const int R0 = g * s.rr, R1 = min(s.H - 1, R0 + s.rr - 1);
    const int lpp = s.lpp, ch = lpp * 8, spw = 64 / lpp;       // lanes per pixel, channels of the strip, pixel slots per wave
    const int slot = lane / lpp, cl = lane - slot * lpp;
    const int W = s.W, rb = s.rb;
    const int n_bands = (R1 - R0 + rb) / rb;
    const uint16_t* const gbase = s.data + (long long)img * p.img_stride[si] + (size_t)chunk_id * ch;
    const int band_elems = rb * W * lpp;                        // 16-byte elements of a band (<= kHfreBandBuf / 16)

    for (int scan_from = 0; scan_from < p.n_boxes;) {
        // ---- the boxes of this image that touch the row range, in box order (ordered compaction), at most kHfreBandMaxBoxes ----
        if (tid == 0) { s_total = 0; s_next = p.n_boxes; }
        __syncthreads();
        for (int base = scan_from; base < p.n_boxes; base += kHfreBandThreads) {
            const int n = base + tid;
            bool m = false;
            int r_lo = 0, r_hi = -1, c_lo = 0, c_hi = -1;
            if (n < p.n_boxes && (p.box_image ? p.box_image[n] : 0) == img) {
                const int* h = p.hdr + ((size_t)n * p.n_sources + si) * 8;
                r_lo = h[0]; r_hi = h[1]; c_lo = h[2]; c_hi = h[3];
                m = h[5] > 0 && r_lo <= R1 && r_hi >= R0;
            }
            const unsigned long long mask = __ballot(m);
            if (lane == 0) s_wcnt[wave] = __popcll(mask);
            __syncthreads();
            int off = s_total;
            for (int w = 0; w < wave; ++w) off += s_wcnt[w];
            if (m) {
                const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < kHfreBandMaxBoxes) { s_box[pos] = n; s_rlo[pos] = r_lo; s_rhi[pos] = r_hi; s_clo[pos] = c_lo; s_chi[pos] = c_hi; }
                else atomicMin(&s_next, n);
            }
            __syncthreads();
            if (tid == 0) {
                int t = s_total;
                for (int w = 0; w < kHfreBandWaves; ++w) t += s_wcnt[w];
                s_total = t;
            }
            __syncthreads();
            if (s_total >= kHfreBandMaxBoxes) {               // uniform: the list is full — whatever was not scanned waits for another pass
                if (tid == 0 && base + kHfreBandThreads < s_next) s_next = base + kHfreBandThreads;
                __syncthreads();
                break;
            }
        }
        const int n_list = min(s_total, kHfreBandMaxBoxes);
        scan_from = s_next;                                    // first box left for another pass (n_boxes: none)
        if (n_list == 0) break;
        for (int i = tid; i < n_list * ch; i += kHfreBandThreads) s_acc[i] = 0.f;

        // ---- stream the strip: band b+1 is loaded while band b is multiplied ----
        hf_u32x4 stage[kHfreBandLoads];
        auto load_band = [&](int b) __attribute__((always_inline)) {
            const int row0 = R0 + b * rb;
#pragma unroll
            for (int i = 0; i < kHfreBandLoads; ++i) {
                int e = tid + i * kHfreBandThreads;
                e = e < band_elems ? e : band_elems - 1;                   // clamped: no load under a branch
                const int pix = e / lpp, q = e - pix * lpp;
                int r = row0 + pix / W;
                r = r <= R1 ? r : R1;
                const int c = pix % W;
                stage[i] = *reinterpret_cast<const hf_u32x4*>(gbase + ((size_t)r * W + c) * s.ld + q * 8);
            }
        };
        auto store_band = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kHfreBandLoads; ++i) {
                const int e = tid + i * kHfreBandThreads;
                if (e < band_elems) *reinterpret_cast<hf_u32x4*>(s_band + buf * kHfreBandBuf + e * 16) = stage[i];
            }
        };
        load_band(0);
        store_band(0);
        __syncthreads();
        for (int b = 0; b < n_bands; ++b) {
            const int br0 = R0 + b * rb, br1 = min(R1, br0 + rb - 1);
            load_band(b + 1 < n_bands ? b + 1 : b);                        // (the last iteration reloads its own band: harmless, never stored)
            const unsigned char* tile = s_band + (b & 1) * kHfreBandBuf;
            for (int li = wave; li < n_list; li += kHfreBandWaves) {
                const int r_lo = s_rlo[li], r_hi = s_rhi[li];
                if (r_hi < br0 || r_lo > br1) continue;                    // wave-uniform
                const int c_lo = s_clo[li], c_hi = s_chi[li], n = s_box[li];
                const int ra = max(r_lo, br0), rz = min(r_hi, br1);
                const float* gwy = p.wbuf + ((size_t)n * p.n_sources + si) * 2 * kHfreWStride;
                const float* gwx = gwy + kHfreWStride;
                float acc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                for (int cb = c_lo; cb <= c_hi; cb += 8 * spw) {            // column block: 8 columns per pixel slot
                    float wx[8];
                    int cc[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = cb + slot + u * spw;
                        const bool ok = c <= c_hi;
                        cc[u] = ok ? c : c_hi;
                        wx[u] = gwx[cc[u] - c_lo];
                        wx[u] = ok ? wx[u] : 0.f;
                    }
                    for (int r = ra; r <= rz; ++r) {
                        const float wy = gwy[r - r_lo];
                        const unsigned char* row = tile + ((size_t)(r - br0) * W) * (lpp * 16) + cl * 16;
                        uint4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(row + (size_t)cc[u] * (lpp * 16));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const float w = wy * wx[u];
                            acc[0] = fmaf(w, bf16_lo(v[u].x), acc[0]);
                            acc[1] = fmaf(w, bf16_hi(v[u].x), acc[1]);
                            acc[2] = fmaf(w, bf16_lo(v[u].y), acc[2]);
                            acc[3] = fmaf(w, bf16_hi(v[u].y), acc[3]);
                            acc[4] = fmaf(w, bf16_lo(v[u].z), acc[4]);
                            acc[5] = fmaf(w, bf16_hi(v[u].z), acc[5]);
                            acc[6] = fmaf(w, bf16_lo(v[u].w), acc[6]);
                            acc[7] = fmaf(w, bf16_hi(v[u].w), acc[7]);
                        }
                    }
                }
                for (int off = 32; off >= lpp; off >>= 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
                }
                if (lane < lpp) {                                           // slot 0 holds the band's sum: this wave owns the box's row
                    float* a = s_acc + li * ch + cl * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] += acc[j];
                }
            }
            store_band((b + 1) & 1);       // last read one band ago: every wave has passed the barrier since
            __syncthreads();
        }
        // ---- the range's partial rows: slice k of box n = range g - first range of the box ----
        const int q4 = ch / 4;
        for (int i = tid; i < n_list * q4; i += kHfreBandThreads) {
            const int li = i / q4, j = i - li * q4;
            const int n = s_box[li], k = g - s_rlo[li] / s.rr;
            float* dst = p.ws + (size_t)n * p.ws_box_stride + s.ws_off + (size_t)k * s.C + (size_t)chunk_id * ch + j * 4;
            *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(s_acc + li * ch + j * 4);
        }
        __syncthreads();                   // s_acc / lists are rewritten by the next pass
    }
}

#endif   // FO1_ENABLE_AB (band form)

// grid (n_boxes, nseg): nseg channel segments per row without LayerNorm, 1 with (the statistics need the whole block)
__global__ __launch_bounds__(kHfreThreads) void hfre_finish2_kernel(const HfreParams p) {
    __shared__ int s_misc[FO1_HFRE_MAX_SOURCES + 4];
    __shared__ float s_stat[8];
    hfre_finish_row(p, blockIdx.x, blockIdx.y, gridDim.y, s_misc, s_stat);
    // every list walker has finished before this kernel starts: leave the item counter at zero for the next call
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kHfreBuckets) p.n_items[threadIdx.x * kHfreCtrStride] = 0;
}

// Vector form of the finish without LayerNorm: a thread owns 4 consecutive output channels (one 16-byte load per slice, the loads of 4
// slices in flight, one 16-byte store), grid (n_boxes, ceil(region_dim / 1024)); same additions in the same order as
// hfre_finish_row, dim_t from the table hfre_weights_kernel wrote.  Needs region_dim % 16 == 0 and 4-aligned source offsets.
__global__ __launch_bounds__(kHfreThreads) void hfre_finish_vec_kernel(const HfreParams p) {
    __shared__ int s_nsl[FO1_HFRE_MAX_SOURCES];
    __shared__ float s_box[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (n == 0 && blockIdx.y == 0 && tid < kHfreBuckets) p.n_items[tid * kHfreCtrStride] = 0;   // every list walker has finished
    if (tid < p.n_sources) s_nsl[tid] = p.hdr[((size_t)n * p.n_sources + tid) * 8 + 5];
    if (tid == 32 && p.pos_mode != 0) {
        float x1, y1, x2, y2;
        load_box(p, n, p.pos_mode == 1, x1, y1, x2, y2);
        x1 = x1 / p.pos_w; x2 = x2 / p.pos_w;       // reference :457-463 — normalise, xyxy -> cxcywh (fp32, same op order)
        y1 = y1 / p.pos_h; y2 = y2 / p.pos_h;
        const float w = x2 - x1, h = y2 - y1;
        s_box[0] = x1 + w / 2.0f;
        s_box[1] = y1 + h / 2.0f;
        s_box[2] = w;
        s_box[3] = h;
    }
    __syncthreads();
    const int c = (blockIdx.y * kHfreThreads + tid) * 4;
    if (c >= p.region_dim) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < p.n_sources; ++i) {
        const HfreSrcDev& q = p.src[i];
        if (c >= q.out_offset && c < q.out_offset + q.C) {
            const float* w = p.ws + (size_t)n * p.ws_box_stride + q.ws_off + (c - q.out_offset);
            const int nsl = s_nsl[i];
            for (int k0 = 0; k0 < nsl; k0 += 4) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kc = (k0 + u < nsl) ? k0 + u : nsl - 1;      // clamped address: no load under a branch
                    t[u] = *reinterpret_cast<const float4*>(w + (size_t)kc * q.C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = k0 + u < nsl;
                    v.x = ok ? v.x + t[u].x : v.x; v.y = ok ? v.y + t[u].y : v.y;
                    v.z = ok ? v.z + t[u].z : v.z; v.w = ok ? v.w + t[u].w : v.w;
                }
            }
        }
    }
    if (p.pos_mode != 0) {
        const int d = p.region_dim / 4;
        const int qd = c / d, i = c - qd * d;                // i % 4 == 0: channels (i, i+1) share dim_t[i/2], (i+2, i+3) dim_t[i/2+1]
        const float coord = (qd == 0) ? s_box[1] : (qd == 1) ? s_box[0] : s_box[qd];
        const float2 dt = *reinterpret_cast<const float2*>(p.dimt + (i >> 1));
        const float a0 = coord * 6.283185307179586f / dt.x, a1 = coord * 6.283185307179586f / dt.y;
        v.x += sinf(a0); v.y += cosf(a0); v.z += sinf(a1); v.w += cosf(a1);
    }
    *reinterpret_cast<float4*>(p.out + (size_t)n * p.out_ld + c) = v;
    if (p.out_bf16) {
        uint2 b;
        b.x = pack_bf16x2(v.x, v.y);
        b.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(p.out_bf16 + (size_t)n * p.out_bf16_ld + c) = b;
    }
}

FO1_AB_VAR g_hfre_pixel_budget = 0;  // 0 = auto
// work-list path
FO1_AB_VAR g_hfre_unroll = 8;        // independent 16-B loads per lane in flight (8 or 16)
FO1_AB_VAR g_hfre_chunk = kHfreMaxChunk;   // channels per workgroup (<= kHfreMaxChunk)
FO1_AB_VAR g_hfre_v2_budget = 256;   // pixels per slice: one value for every box count, so a box's result does not depend on what
                                     // else is in the call (batch invariance)
FO1_AB_VAR g_hfre_finish_vec = 1;     // 16-byte finish (A/B: fo1_hfre_set_tuning unroll | 32 turns it off)
FO1_AB_VAR g_hfre_grid = 4096;       // workgroups walking the work list (profiles/r02_hfre_sweep.json)
FO1_AB_VAR g_hfre_bands = 0;         // 0 = work list (default), 1 = band kernel (every map row read once: measured 4x slower, A/B only); fo1_hfre_set_tuning(budget = -1 / -2)
FO1_AB_VAR g_hfre_rr = 32;           // band path: rows per work item
FO1_AB_VAR g_hfre_order = 1;         // work-list order: 1 = box-major (image-major) buckets, 0 = pairs interleaved over the 64 buckets (rounds 2-5); fo1_hfre_set_tuning(budget = -4 / -3)

// workspace = [partials: n_boxes * ws_box_stride floats][weights: n_boxes*n_sources*2*kHfreWStride floats][headers: n_boxes*n_sources*8 ints]
static size_t hfre_ws_total(const HfreParams& p, int n_sources, int n_boxes) {
    const size_t nb = (size_t)(n_boxes > 0 ? n_boxes : 1);
    return ((size_t)p.ws_box_stride * nb + nb * n_sources * 2 * kHfreWStride) * sizeof(float) + nb * n_sources * 8 * sizeof(int);
}

static int hfre_plan(const fo1_hfre_source_t* sources, int n_sources, int n_boxes, HfreParams& p, int& total_wgs, bool v2 = false, int n_images = 1) {
    FO1_CHECK_ARG(sources != nullptr, "hfre: sources is NULL");
    FO1_CHECK_ARG(n_sources >= 1 && n_sources <= FO1_HFRE_MAX_SOURCES, "hfre: n_sources=%d out of [1,%d]", n_sources,
                  FO1_HFRE_MAX_SOURCES);
    FO1_CHECK_ARG(n_boxes >= 0, "hfre: n_boxes=%d < 0", n_boxes);
    p.n_sources = n_sources;
    p.band_mode = (v2 && g_hfre_bands) ? 1 : 0;
    p.n_images = n_images > 0 ? n_images : 1;
    // measured on MI355X (profiles/r01_hfre.md): per-workgroup streaming is latency-bound (~16 KB in flight), so
    // small slices win until the empty-workgroup count takes over: 256 px up to ~48 boxes, 512 px beyond
    p.pixel_budget = g_hfre_pixel_budget > 0 ? g_hfre_pixel_budget : (v2 ? g_hfre_v2_budget : (n_boxes <= 48 ? 256 : 512));
    int wg = 0, ws = 0;
    for (int i = 0; i < n_sources; ++i) {
        const fo1_hfre_source_t& s = sources[i];
        FO1_CHECK_ARG(s.H >= 1 && s.W >= 1 && s.H <= FO1_HFRE_MAX_EXTENT && s.W <= FO1_HFRE_MAX_EXTENT,
                      "hfre: source %d map %dx%d outside [1,%d]", i, s.H, s.W, FO1_HFRE_MAX_EXTENT);
        FO1_CHECK_ARG(s.roi_H >= s.H && s.roi_W >= s.W && s.roi_H <= FO1_HFRE_MAX_EXTENT && s.roi_W <= FO1_HFRE_MAX_EXTENT,
                      "hfre: source %d roi map %dx%d must be >= map %dx%d and <= %d", i, s.roi_H, s.roi_W, s.H, s.W,
                      FO1_HFRE_MAX_EXTENT);
        FO1_CHECK_ARG(s.C >= 64 && s.C % 64 == 0, "hfre: source %d C=%d must be a positive multiple of 64", i, s.C);
        FO1_CHECK_ARG(s.ld >= s.C && s.ld % 8 == 0, "hfre: source %d ld=%d must be >= C and a multiple of 8", i, s.ld);
        FO1_CHECK_ARG(s.box_space == 0 || s.box_space == 1, "hfre: source %d box_space=%d", i, s.box_space);
        FO1_CHECK_ARG(s.out_offset >= 0, "hfre: source %d out_offset=%d", i, s.out_offset);
        HfreSrcDev& d = p.src[i];
        d.data = (const uint16_t*)s.data;
        d.H = s.H; d.W = s.W; d.C = s.C; d.ld = s.ld; d.roi_H = s.roi_H; d.roi_W = s.roi_W;
        d.scale = s.spatial_scale; d.box_space = s.box_space; d.out_offset = s.out_offset;
        int chunk = v2 ? g_hfre_chunk : kHfreMaxChunk;
        while (s.C % chunk) chunk >>= 1;
        d.chunk = chunk;
        d.nchunks = s.C / chunk;
        d.max_slices = cdiv(s.H, slice_rows(s.W, p.pixel_budget));
        // band path: the widest channel strip whose single row fits an LDS band buffer, as many rows per band as fit, rr rows per item
        int lpp = 8;
        while (lpp > 1 && (long long)s.W * lpp * 16 > 40960) lpp >>= 1;
        d.lpp = lpp;
        d.rb = (int)(40960 / ((long long)s.W * lpp * 16));
        if (d.rb < 1) d.rb = 1;                      // (W <= FO1_HFRE_MAX_EXTENT = 1024 always fits at lpp = 2)
        if (d.rb > 16) d.rb = 16;
        d.rr = g_hfre_rr > d.rb ? g_hfre_rr : d.rb;
        d.n_ranges = cdiv(s.H, d.rr);
        d.bchunks = s.C / (lpp * 8);
        d.item_base = 0;
        if (p.band_mode) d.max_slices = d.n_ranges;
        d.wg_base = wg;
        d.ws_off = ws;
        wg += n_boxes * d.nchunks * d.max_slices;
        ws += d.max_slices * s.C;
    }
    p.ws_box_stride = ws;
    total_wgs = wg;
    // band items: the largest strips first (one workgroup per item, dispatched in order: the long ones must not start last)
    p.band_items = 0;
    bool placed[FO1_HFRE_MAX_SOURCES] = {false};
    for (int k = 0; k < n_sources; ++k) {
        int best = -1;
        long long best_sz = -1;
        for (int i = 0; i < n_sources; ++i) {
            const long long sz = (long long)(p.src[i].rr < p.src[i].H ? p.src[i].rr : p.src[i].H) * p.src[i].W * p.src[i].lpp;
            if (!placed[i] && sz > best_sz) { best = i; best_sz = sz; }
        }
        placed[best] = true;
        p.src[best].item_base = p.band_items;
        p.band_items += p.n_images * p.src[best].bchunks * p.src[best].n_ranges;
    }
    return FO1_OK;
}

}  // namespace fo1

extern "C" {

#ifdef FO1_ENABLE_AB      // include/fo1_ab.h: test / bench build only
// test/tuning hook: pixels per workgroup slice (default 1024)
int fo1_hfre_set_pixel_budget(int pixels) {
    if (pixels != 0 && (pixels < 16 || pixels > 65536)) return fo1::set_err(FO1_ERR_ARG, "hfre: pixel budget %d outside [16,65536] (0 = auto)", pixels);
    fo1::g_hfre_pixel_budget = pixels;
    return FO1_OK;
}

// tuning hooks of fo1_hfre_region_pool_ex: unroll 8 | 16 independent loads per lane; chunk = channels per workgroup (64..512, power
// of two); budget = pixels per slice (0 keeps the current value); grid = workgroups walking the work list (0 keeps)
int fo1_hfre_set_tuning(int unroll, int chunk, int budget, int grid) {
    if (budget == -3 || budget == -4) {                // A/B: work-list order, -3 = interleaved buckets (rounds 2-5), -4 = box-major buckets (default)
        fo1::g_hfre_order = budget == -4 ? 1 : 0;
        return FO1_OK;
    }
    if (budget == -1 || budget == -2) {                // A/B: -1 = work-list form (default), -2 = band form; the rest is ignored
        fo1::g_hfre_bands = budget == -2 ? 1 : 0;
        if (grid > 0) fo1::g_hfre_rr = grid;           // with -2: rows per work item
        return FO1_OK;
    }
    fo1::g_hfre_finish_vec = (unroll & 32) ? 0 : 1;    // A/B: unroll | 32 = scalar finish
    unroll &= ~32;
    if ((unroll != 8 && unroll != 16) || chunk < 64 || chunk > fo1::kHfreMaxChunk || (chunk & (chunk - 1)) ||
        (budget != 0 && (budget < 16 || budget > 65536)) || grid < 0 || grid > (1 << 20))
        return fo1::set_err(FO1_ERR_ARG, "hfre: set_tuning(unroll=%d, chunk=%d, budget=%d, grid=%d)", unroll, chunk, budget, grid);
    fo1::g_hfre_unroll = unroll; fo1::g_hfre_chunk = chunk;
    if (budget) fo1::g_hfre_v2_budget = budget;
    if (grid) fo1::g_hfre_grid = grid;
    return FO1_OK;
}
#endif   // FO1_ENABLE_AB

#ifdef FO1_ENABLE_AB
size_t fo1_hfre_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes) {
    fo1::HfreParams p;
    int wgs = 0;
    if (fo1::hfre_plan(sources, n_sources, n_boxes, p, wgs) != FO1_OK) return 0;
    return hfre_ws_total(p, n_sources, n_boxes);
}

int fo1_hfre_region_pool(const fo1_hfre_source_t* sources, int n_sources, const float* boxes_aux, int n_boxes,
                         const float* boxes_vt, float vt_scale_x, float vt_scale_y, int roi_size, int pos_mode, float pos_img_w,
                         float pos_img_h, float* out, int out_ld, int region_dim, void* workspace,
                         size_t workspace_bytes, void* stream) {
    using namespace fo1;
    HfreParams p;
    int wgs = 0;
    int rc = hfre_plan(sources, n_sources, n_boxes, p, wgs);
    if (rc != FO1_OK) return rc;
    if (n_boxes == 0) return FO1_OK;
    FO1_CHECK_ARG(boxes_aux != nullptr && out != nullptr, "hfre: NULL boxes/out");
    FO1_CHECK_ARG(((uintptr_t)boxes_aux & 15) == 0 && ((uintptr_t)boxes_vt & 15) == 0,
                  "hfre: boxes must be 16-byte aligned");
    FO1_CHECK_ARG(roi_size >= 1 && roi_size <= 32, "hfre: roi_size=%d", roi_size);
    FO1_CHECK_ARG(pos_mode >= 0 && pos_mode <= 2, "hfre: pos_mode=%d", pos_mode);
    FO1_CHECK_ARG(region_dim >= 4 && out_ld >= region_dim, "hfre: region_dim=%d out_ld=%d", region_dim, out_ld);
    FO1_CHECK_ARG(pos_mode == 0 || (region_dim % 4 == 0 && pos_img_w > 0.f && pos_img_h > 0.f),
                  "hfre: position embedding needs region_dim %% 4 == 0 and positive image size");
    for (int i = 0; i < n_sources; ++i) {
        FO1_CHECK_ARG(sources[i].data != nullptr && ((uintptr_t)sources[i].data & 15) == 0,
                      "hfre: source %d data NULL or not 16-byte aligned", i);
        FO1_CHECK_ARG(sources[i].out_offset + sources[i].C <= region_dim, "hfre: source %d writes past region_dim", i);
    }
    const size_t need = hfre_ws_total(p, n_sources, n_boxes);
    if (workspace == nullptr || workspace_bytes < need)
        return set_err(FO1_ERR_WORKSPACE, "hfre: workspace %zu B < required %zu B", workspace_bytes, need);
    p.boxes = boxes_aux;
    p.boxes_vt = boxes_vt;
    p.n_boxes = n_boxes;
    p.vsx = vt_scale_x;
    p.vsy = vt_scale_y;
    p.P = roi_size;
    p.pos_mode = pos_mode;
    p.pos_w = pos_img_w;
    p.pos_h = pos_img_h;
    p.out = out;
    p.out_ld = out_ld;
    p.out_bf16 = nullptr; p.out_bf16_ld = 0;
    p.region_dim = region_dim;
    p.ws = (float*)workspace;
    p.wbuf = p.ws + (size_t)p.ws_box_stride * n_boxes;
    p.hdr = (int*)(p.wbuf + (size_t)n_boxes * n_sources * 2 * kHfreWStride);
    p.items = nullptr; p.n_items = nullptr; p.box_image = nullptr; p.dimt = nullptr;
    hipStream_t st = (hipStream_t)stream;
    FO1_LAUNCH("hfre_weights", (double)n_boxes * n_sources * 64.0, hfre_weights_kernel, dim3(n_boxes * n_sources), dim3(kHfreWThreads), 0, st, p);
    // algorithmic bytes (SURVEY §8d upper bound): every source map once in bf16 + fp32 output + boxes
    double bytes = (double)n_boxes * region_dim * 4.0 + (double)n_boxes * 16.0;
    for (int i = 0; i < n_sources; ++i) bytes += (double)sources[i].H * sources[i].W * sources[i].C * 2.0;
    FO1_LAUNCH("hfre_pool", bytes, hfre_pool_kernel, dim3(wgs), dim3(kHfreThreads), 0, st, p);
    FO1_LAUNCH("hfre_finish", (double)n_boxes * region_dim * 8.0, hfre_finish_kernel,
               dim3(cdiv(region_dim, 256), n_boxes), dim3(256), 0, st, p);
    return FO1_OK;
}
#else
// Product build: the single-image entry points of round 1 (the name SURVEY 8b gives the HFRE seam) are the work-list path with default
// options.  The work-list counters must be zero on a workspace's first use: this form, whose contract never asked the caller for a
// zeroed workspace, clears them itself (one 64-thread launch).
size_t fo1_hfre_ex_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes);
int fo1_hfre_region_pool_ex(const fo1_hfre_source_t* sources, int n_sources, const float* boxes_aux, int n_boxes, const float* boxes_vt,
                            float vt_scale_x, float vt_scale_y, int roi_size, int pos_mode, float pos_img_w, float pos_img_h, float* out, int out_ld,
                            int region_dim, const fo1_hfre_opts_t* opts, void* workspace, size_t workspace_bytes, void* stream);
size_t fo1_hfre_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes) {
    return fo1_hfre_ex_workspace_bytes(sources, n_sources, n_boxes);
}
namespace fo1 {
__global__ void hfre_zero_counters_kernel(int* ctr) { ctr[threadIdx.x * kHfreCtrStride] = 0; }
}
int fo1_hfre_region_pool(const fo1_hfre_source_t* sources, int n_sources, const float* boxes_aux, int n_boxes,
                         const float* boxes_vt, float vt_scale_x, float vt_scale_y, int roi_size, int pos_mode, float pos_img_w,
                         float pos_img_h, float* out, int out_ld, int region_dim, void* workspace,
                         size_t workspace_bytes, void* stream) {
    using namespace fo1;
    if (n_boxes > 0 && workspace != nullptr && workspace_bytes >= (size_t)kHfreBuckets * kHfreCtrStride * sizeof(int))
        FO1_LAUNCH("hfre_zero_counters", 256.0, hfre_zero_counters_kernel, dim3(1), dim3(kHfreBuckets), 0, (hipStream_t)stream, (int*)workspace);
    return fo1_hfre_region_pool_ex(sources, n_sources, boxes_aux, n_boxes, boxes_vt, vt_scale_x, vt_scale_y, roi_size, pos_mode, pos_img_w,
                                   pos_img_h, out, out_ld, region_dim, nullptr, workspace, workspace_bytes, stream);
}
#endif   // FO1_ENABLE_AB


// Work-list variant: same contract as fo1_hfre_region_pool, plus
//   * batch: boxes of several same-geometry images in one call — box n belongs to image opts->box_image[n] and source i of image b
//     starts opts->img_stride[i] elements after image b-1's (sources[i].data addresses image 0);
//   * region LayerNorm (mm_apply_region_layer_norm, reference :365-372): fp32 LayerNorm (biased variance) of the channel blocks
//     [0, ln_split) with (ln_w0, ln_b0) and [ln_split, region_dim) with (ln_w1, ln_b1) BEFORE the position embedding;
//     ln_split = 0 / region_dim means one block (vt-only / aux-only configurations).
// workspace = [16 KB: 64 item counters, 256 B apart][partials][tap weights][headers][work items: 64 buckets].  The counters must be ZERO
// when a workspace is first used (the host module's scratch pool zero-fills on allocation); the finish kernel leaves them at zero for
// the next call.
static constexpr size_t kHfreExCtr = (size_t)fo1::kHfreBuckets * fo1::kHfreCtrStride * sizeof(int);
static constexpr size_t kHfreExHead = kHfreExCtr + 16384;     // + dim_t table (region_dim / 8 floats: region_dim <= 32768)
static int hfre_bucket_cap(const fo1::HfreParams& p, int n_sources, int n_boxes) {
    int max_item = 1;
    for (int i = 0; i < n_sources; ++i) {
        const int m = p.src[i].nchunks * p.src[i].max_slices;
        if (m > max_item) max_item = m;
    }
    const long long pairs = (long long)(n_boxes > 0 ? n_boxes : 1) * n_sources;
    return (int)((pairs + fo1::kHfreBuckets - 1) / fo1::kHfreBuckets) * max_item;
}
static size_t hfre_ex_layout(const fo1::HfreParams& p, int n_sources, int n_boxes, int total_wgs, size_t* off_w, size_t* off_h, size_t* off_i) {
    const size_t nb = (size_t)(n_boxes > 0 ? n_boxes : 1);
    size_t o = kHfreExHead + (size_t)p.ws_box_stride * nb * sizeof(float);
    if (off_w) *off_w = o;
    o += nb * n_sources * 2 * fo1::kHfreWStride * sizeof(float);
    if (off_h) *off_h = o;
    o += nb * n_sources * 8 * sizeof(int);
    if (off_i) *off_i = o;
    (void)total_wgs;
    o += (size_t)fo1::kHfreBuckets * hfre_bucket_cap(p, n_sources, n_boxes) * sizeof(int2);
    return o;
}

size_t fo1_hfre_ex_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes) {
    fo1::HfreParams p;
    int wgs = 0;
    if (fo1::hfre_plan(sources, n_sources, n_boxes, p, wgs, true) != FO1_OK) return 0;
    return hfre_ex_layout(p, n_sources, n_boxes, wgs, nullptr, nullptr, nullptr);
}

int fo1_hfre_region_pool_ex(const fo1_hfre_source_t* sources, int n_sources, const float* boxes_aux, int n_boxes,
                            const float* boxes_vt, float vt_scale_x, float vt_scale_y, int roi_size, int pos_mode, float pos_img_w,
                            float pos_img_h, float* out, int out_ld, int region_dim, const fo1_hfre_opts_t* opts, void* workspace,
                            size_t workspace_bytes, void* stream) {
    using namespace fo1;
    HfreParams p;
    int wgs = 0;
    int rc = hfre_plan(sources, n_sources, n_boxes, p, wgs, true, opts ? opts->batch : 1);
    if (rc != FO1_OK) return rc;
    if (n_boxes == 0) return FO1_OK;
    FO1_CHECK_ARG(boxes_aux != nullptr && out != nullptr, "hfre: NULL boxes/out");
    FO1_CHECK_ARG(((uintptr_t)boxes_aux & 15) == 0 && ((uintptr_t)boxes_vt & 15) == 0, "hfre: boxes must be 16-byte aligned");
    FO1_CHECK_ARG(roi_size >= 1 && roi_size <= 32, "hfre: roi_size=%d", roi_size);
    FO1_CHECK_ARG(pos_mode >= 0 && pos_mode <= 2, "hfre: pos_mode=%d", pos_mode);
    FO1_CHECK_ARG(region_dim >= 4 && out_ld >= region_dim, "hfre: region_dim=%d out_ld=%d", region_dim, out_ld);
    FO1_CHECK_ARG(pos_mode == 0 || (region_dim % 4 == 0 && pos_img_w > 0.f && pos_img_h > 0.f),
                  "hfre: position embedding needs region_dim %% 4 == 0 and positive image size");
    for (int i = 0; i < n_sources; ++i) {
        FO1_CHECK_ARG(sources[i].data != nullptr && ((uintptr_t)sources[i].data & 15) == 0, "hfre: source %d data NULL or not 16-byte aligned", i);
        FO1_CHECK_ARG(sources[i].out_offset + sources[i].C <= region_dim, "hfre: source %d writes past region_dim", i);
        FO1_CHECK_ARG(p.src[i].nchunks <= 0xFFF && p.src[i].max_slices <= 0xFFF, "hfre: source %d has too many chunks / slices for a work item", i);
        p.img_stride[i] = 0;
    }
    p.box_image = nullptr;
    p.ln_on = 0; p.ln_split = 0; p.ln_w0 = p.ln_b0 = p.ln_w1 = p.ln_b1 = nullptr; p.ln_eps = 1e-5f;
    p.out_bf16 = nullptr; p.out_bf16_ld = 0;
    if (opts) {
        FO1_CHECK_ARG(opts->batch >= 1, "hfre: batch=%d", opts->batch);
        FO1_CHECK_ARG(opts->batch == 1 || opts->box_image != nullptr, "hfre: a batched call needs box_image");
        p.box_image = opts->box_image;
        for (int i = 0; i < n_sources; ++i) {
            FO1_CHECK_ARG(opts->img_stride[i] % 8 == 0, "hfre: img_stride[%d] must keep 16-byte alignment", i);
            p.img_stride[i] = opts->img_stride[i];
        }
        if (opts->out_bf16) {
            FO1_CHECK_ARG(opts->out_bf16_ld >= region_dim && opts->out_bf16_ld % 4 == 0 && ((uintptr_t)opts->out_bf16 & 7) == 0,
                          "hfre: out_bf16 needs ld >= region_dim, ld %% 4 == 0 and 8-byte alignment");
            p.out_bf16 = (uint16_t*)opts->out_bf16; p.out_bf16_ld = opts->out_bf16_ld;
        }
        if (opts->ln_on) {
            FO1_CHECK_ARG(opts->ln_split >= 0 && opts->ln_split <= region_dim, "hfre: ln_split=%d", opts->ln_split);
            FO1_CHECK_ARG((opts->ln_split == 0 || (opts->ln_w0 && opts->ln_b0)) && (opts->ln_split == region_dim || (opts->ln_w1 && opts->ln_b1)),
                          "hfre: region LayerNorm needs weight and bias for every block");
            p.ln_on = 1; p.ln_split = opts->ln_split; p.ln_eps = opts->ln_eps;
            p.ln_w0 = opts->ln_w0; p.ln_b0 = opts->ln_b0; p.ln_w1 = opts->ln_w1; p.ln_b1 = opts->ln_b1;
        }
    }
    size_t off_w = 0, off_h = 0, off_i = 0;
    const size_t need = hfre_ex_layout(p, n_sources, n_boxes, wgs, &off_w, &off_h, &off_i);
    if (workspace == nullptr || workspace_bytes < need) return set_err(FO1_ERR_WORKSPACE, "hfre: workspace %zu B < required %zu B", workspace_bytes, need);
    FO1_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "hfre: workspace must be 16-byte aligned");
    p.boxes = boxes_aux; p.boxes_vt = boxes_vt; p.n_boxes = n_boxes; p.vsx = vt_scale_x; p.vsy = vt_scale_y;
    p.P = roi_size; p.pos_mode = pos_mode; p.pos_w = pos_img_w; p.pos_h = pos_img_h;
    p.out = out; p.out_ld = out_ld; p.region_dim = region_dim;
    char* wsb = (char*)workspace;
    p.n_items = (int*)wsb;
    p.ws = (float*)(wsb + kHfreExHead);
    p.dimt = (float*)(wsb + kHfreExCtr);
    p.wbuf = (float*)(wsb + off_w);
    p.hdr = (int*)(wsb + off_h);
    p.items = p.band_mode ? nullptr : (int2*)(wsb + off_i);      // band form: no work list (hfre_weights_kernel skips it)
    p.items_cap = hfre_bucket_cap(p, n_sources, n_boxes);
    p.bucket_per = g_hfre_order ? (int)(((long long)n_boxes * n_sources + kHfreBuckets - 1) / kHfreBuckets) : 0;      // (the capacity rule of hfre_bucket_cap: <= ceil(pairs / 64) pairs per bucket either way)
    hipStream_t st = (hipStream_t)stream;
    FO1_LAUNCH("hfre_weights", (double)n_boxes * n_sources * 64.0, hfre_weights_kernel, dim3(n_boxes * n_sources), dim3(kHfreWThreads), 0, st, p);
    double bytes = (double)n_boxes * region_dim * 4.0 + (double)n_boxes * 16.0;
    for (int i = 0; i < n_sources; ++i) bytes += (double)sources[i].H * sources[i].W * sources[i].C * 2.0 * (opts ? opts->batch : 1);
#ifdef FO1_ENABLE_AB
    if (p.band_mode) {
        static bool attr = false;
        if (!attr) {
            FO1_CHECK_HIP(hipFuncSetAttribute((const void*)hfre_pool_bands_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kHfreBandSmem));
            attr = true;
        }
        FO1_LAUNCH("hfre_pool_bands", bytes, hfre_pool_bands_kernel, dim3(p.band_items), dim3(kHfreBandThreads), kHfreBandSmem, st, p);
    } else
#endif
    {
        const int grid = wgs < g_hfre_grid ? wgs : g_hfre_grid;
        if (g_hfre_unroll == 16) { FO1_LAUNCH("hfre_pool_items", bytes, (hfre_pool_items_kernel<16>), dim3(grid), dim3(kHfreThreads), 0, st, p); }
        else                     { FO1_LAUNCH("hfre_pool_items", bytes, (hfre_pool_items_kernel<8>), dim3(grid), dim3(kHfreThreads), 0, st, p); }
    }
    bool vec = !p.ln_on && region_dim % 16 == 0 && region_dim <= 32768 && out_ld % 4 == 0 && ((uintptr_t)out & 15) == 0 && g_hfre_finish_vec;
    for (int i = 0; i < n_sources; ++i) vec = vec && sources[i].out_offset % 4 == 0;
    if (vec) {
        FO1_LAUNCH("hfre_finish2", (double)n_boxes * region_dim * 8.0, hfre_finish_vec_kernel, dim3(n_boxes, cdiv(region_dim, 4 * kHfreThreads)),
                   dim3(kHfreThreads), 0, st, p);
    } else {
        const int nseg = p.ln_on ? 1 : 4;
        FO1_LAUNCH("hfre_finish2", (double)n_boxes * region_dim * 8.0, hfre_finish2_kernel, dim3(n_boxes, nseg), dim3(kHfreThreads), 0, st, p);
    }
    return FO1_OK;
}

}  // extern "C"
