// vision_ops.hip — data-movement and small-reduction kernels of the DaViT / SimpleFPN / splice
// stages (all HBM-bound, 16-byte bf16x8 accesses on token-major [H*W, C] maps):
//   dwconv3x3      DepthWiseConv2d + PreNorm(None) residual          modeling_davit.py:72-99,29-48
//   im2col         ConvEmbed / 3x3 FPN conv as implicit-GEMM staging modeling_davit.py:102-148, simple_fpn.py:141-176
//   window part./reverse  WindowAttention pad+partition / merge+crop  modeling_davit.py:208-222,244-254,272-281
//   channel attention     ChannelAttention                            modeling_davit.py:151-172
//   pixel shuffle   ConvTranspose2d(k=2,s=2) epilogue                 simple_fpn.py:141-150
//   maxpool2x2      nn.MaxPool2d(2,2)                                 simple_fpn.py:153
//   nchw_to_hwc8    image [3,H,W] -> token-major with 8 channels (conv-embed staging)
//   gather_rows     embedding lookup + image/region token splice      omchat_qwen2_5_vl.py:291-373
#include "common.h"

namespace fo1 {

__device__ __forceinline__ void un8(const uint4& u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pk8(const float (&f)[8]) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    return u;
}
__device__ __forceinline__ float rbf(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// Ragged image batches (images of DIFFERENT sizes packed row-wise into one map, include/fo1.h fo1_img_seg): when `segs` is given,
// workgroup column blockIdx.y works on image blockIdx.y alone — its own H x W, its rows starting at in_row0 / out_row0 — with exactly
// the per-image arithmetic of the same-size path (B = 1), so a packed ragged pass is bit-identical to the one-image passes.
struct ImgSeg { int in_row0, H, W, out_row0, Ho, Wo, a0, a1; };

// y[h,w,c] = x[h,w,c] + bf16( sum_{ky,kx} x[h+ky-1, w+kx-1, c] * wt[(ky*3+kx)*C + c] + bias[c] )
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                        const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int H, int W, int C, int B) {
    const int chunks = C >> 3;
    const int HW = H * W;
    const long long total = (long long)B * HW * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int pix = (int)(i / chunks);           // global pixel over the B images
        const int img0 = (pix / HW) * HW, lp = pix - img0;
        const int h = lp / W, w = lp - h * W;
        float acc[8], ctr[8];
        un8(*reinterpret_cast<const uint4*>(bias + c * 8), acc);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hh = h + ky - 1;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ww = w + kx - 1;
                if (ww < 0 || ww >= W) continue;
                float xv[8], wv[8];
                un8(*reinterpret_cast<const uint4*>(x + ((long long)img0 + hh * W + ww) * C + c * 8), xv);
                un8(*reinterpret_cast<const uint4*>(wt + (ky * 3 + kx) * C + c * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
                if (ky == 1 && kx == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ctr[j] = xv[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = ctr[j] + rbf(acc[j]);
        *reinterpret_cast<uint4*>(y + (long long)pix * C + c * 8) = pk8(acc);
    }
}

// Depthwise 3x3 + residual AND the LayerNorm every DaViT block applies right after it, one launch:
//   y = x + bf16(dwconv(x) + bias)            (the residual stream, as dwconv3x3_kernel)
//   h = LayerNorm(y) * ln_w + ln_b            (as rownorm_kernel<1>: two-pass variance from registers)
// One wave per pixel: lane l owns channel chunks l, l+64, ... (8 channels each), so the taps are 1 KB-contiguous reads and the
// row statistics are one wave reduction.  Arithmetic order matches the two separate kernels bit for bit.
constexpr int kDwLnChunks = 4;   // C <= 64 * 4 * 8 = 2048
__device__ __forceinline__ float dw_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// PPW pixels per wave (C = 256 fills only half a wave's lanes with one pixel).  All 18 loads of a chunk are issued together: taps outside
// the map read a clamped (valid) address and meet a zero weight — fmaf(x, 0, acc) leaves acc as it was, so the sums are the ones the
// branchy form produced (a load under a branch costs a vmcnt(0) at the join: nine dependent L2 round trips per pixel, 3 TB/s).
template <int PPW>
__global__ __launch_bounds__(256) void dwconv3x3_ln_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                           const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
                                                           const uint16_t* __restrict__ ln_w, const uint16_t* __restrict__ ln_b,
                                                           uint16_t* __restrict__ hout, int H, int W, int C, float eps, int B,
                                                           const ImgSeg* __restrict__ segs) {
    constexpr int LPP = 64 / PPW;                              // lanes per pixel
    const int lane = threadIdx.x & 63, sub = lane % LPP;
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; B = 1;
        x += (long long)sg.in_row0 * C; y += (long long)sg.in_row0 * C; hout += (long long)sg.in_row0 * C;
        if ((long long)blockIdx.x * 4 * PPW >= (long long)H * W) return;     // whole workgroup past this image (uniform: no shuffle is split)
    }
    const int HW = H * W;
    int pix = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;      // global pixel over the B images
    const bool live = pix < B * HW;
    if (!live) pix = B * HW - 1;                               // (keeps the half-wave shuffles well defined; nothing is stored)
    const int chunks = C >> 3;
    const int img0 = (pix / HW) * HW, lp = pix - img0;
    const int h = lp / W, w = lp - h * W;
    float val[kDwLnChunks][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kDwLnChunks; ++i) {
        const int c = sub + i * LPP;
        if (c < chunks) {
            float acc[8], ctr[8];
            un8(*reinterpret_cast<const uint4*>(bias + c * 8), acc);
            uint4 xr[9], wr[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int hh = h + ky - 1, ww = w + kx - 1;
                    const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                    const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh), wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
                    xr[ky * 3 + kx] = *reinterpret_cast<const uint4*>(x + ((long long)img0 + hc * W + wc) * C + c * 8);
                    const uint4 wq = *reinterpret_cast<const uint4*>(wt + (ky * 3 + kx) * C + c * 8);
                    wr[ky * 3 + kx] = ok ? wq : uint4{0, 0, 0, 0};
                }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float xv[8], wv[8];
                un8(xr[t], xv);
                un8(wr[t], wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
                if (t == 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ctr[j] = xv[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = ctr[j] + rbf(acc[j]);
            const uint4 packed = pk8(acc);
            if (live) *reinterpret_cast<uint4*>(y + (long long)pix * C + c * 8) = packed;
            un8(packed, val[i]);              // LayerNorm sees the bf16 values that were stored
#pragma unroll
            for (int j = 0; j < 8; ++j) s += val[i][j];
        }
    }
    auto psum = [](float v) __attribute__((always_inline)) {     // over the LPP lanes of this pixel (the 64-lane tree minus its zero halves)
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    s = psum(s);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kDwLnChunks; ++i) {
        const int c = sub + i * LPP;
        if (c < chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = val[i][j] - mean; q = fmaf(d, d, q); }       // explicit fmaf chains: as rownorm_kernel<1>
        }
    }
    q = psum(q);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < kDwLnChunks; ++i) {
        const int c = sub + i * LPP;
        if (c < chunks && live) {
            float wf[8], bf[8], o[8];
            un8(*reinterpret_cast<const uint4*>(ln_w + c * 8), wf);
            un8(*reinterpret_cast<const uint4*>(ln_b + c * 8), bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf((val[i][j] - mean) * rstd, wf[j], bf[j]);
            *reinterpret_cast<uint4*>(hout + (long long)pix * C + c * 8) = pk8(o);
        }
    }
}

// Round 6: the same arithmetic with a SLIDING WINDOW — a wave walks a run of kDwRun consecutive pixels of one image row.  The 3 x 3 window of
// every channel chunk lives in registers (a ring of four columns: three in use, the fourth receiving the next column while the taps run), so a
// pixel costs 3 tap loads instead of 9, and the nine weight vectors, the bias and the LayerNorm parameters are loaded once per run instead of
// once per pixel (18 + 2 loads per pixel and chunk -> 3.75): the per-pixel kernel kept the CU's texture path busy 2.6x longer than HBM needs.
// Fully unrolled (static ring indices, no loop back-edge: a back-edge costs a vmcnt(0) that would also wait for the stores).  Rows outside the
// image zero the row's weights once per run, columns outside zero the tap's weight at the two pixels that see them: fmaf(x, 0, acc) = acc,
// the per-pixel kernel's rule, same sums bit for bit (tests/test_vision_ops_gpu.py).  Needs C = 8 * NCH * 64 / PPW exactly.
constexpr int kDwRun = 8;
template <int PPW, int NCH, int RUN>
__global__ __launch_bounds__(256) void dwconv3x3_ln_run_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                               const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
                                                               const uint16_t* __restrict__ ln_w, const uint16_t* __restrict__ ln_b,
                                                               uint16_t* __restrict__ hout, int H, int W, int C, float eps, int B,
                                                               const ImgSeg* __restrict__ segs) {
    constexpr int LPP = 64 / PPW;
    const int lane = threadIdx.x & 63, sub = lane % LPP;
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; B = 1;
        x += (long long)sg.in_row0 * C; y += (long long)sg.in_row0 * C; hout += (long long)sg.in_row0 * C;
    }
    const int rpr = (W + RUN - 1) / RUN, runs_img = H * rpr;
    if (segs && (long long)blockIdx.x * 4 * PPW >= runs_img) return;            // whole workgroup past this image (uniform)
    int run = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const bool run_live = run < B * runs_img;
    if (!run_live) run = B * runs_img - 1;                                       // (keeps the shuffles well defined; nothing is stored)
    const int img = run / runs_img, rr = run - img * runs_img;
    const int h = rr / rpr, w0 = (rr - h * rpr) * RUN;
    const long long img_pix0 = (long long)img * H * W;
    // stores through buffer descriptors: a pixel past the row's end (or a run past the last) stores at an offset outside the descriptor and the
    // hardware drops it — no branch around a store, so the wait for a prefetched column never has to assume the stores in between were skipped
    // (hipcc's waitcnt pass would then wait for the stores themselves: one store round trip per pixel).  Map <= 2 GiB (host-checked).
    typedef __attribute__((ext_vector_type(4))) unsigned int v4u;
    const uint32_t map_bytes = (uint32_t)B * (uint32_t)H * (uint32_t)W * (uint32_t)C * 2u;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, map_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)hout, 0, map_bytes, 0x00020000);
    const uint32_t row_off = ((uint32_t)img_pix0 + (uint32_t)h * (uint32_t)W) * (uint32_t)C * 2u + (uint32_t)sub * 16u;
    const bool rok[3] = {h > 0, true, h + 1 < H};
    const uint16_t* const rowp[3] = {x + (img_pix0 + (long long)(h > 0 ? h - 1 : 0) * W) * C + sub * 8, x + (img_pix0 + (long long)h * W) * C + sub * 8,
                                     x + (img_pix0 + (long long)(h + 1 < H ? h + 1 : H - 1) * W) * C + sub * 8};
    auto colc = [&](int wcol) { return wcol < 0 ? 0 : (wcol >= W ? W - 1 : wcol); };
    auto psum = [](float v) __attribute__((always_inline)) {
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const uint4 zero4 = uint4{0, 0, 0, 0};
    // phase 1, chunk by chunk: the convolution of the run's pixels (y stored, its bf16 values kept packed: 4 registers per pixel and chunk)
    uint4 pk[NCH][RUN];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int co = (sub + i * LPP) * 8;
        uint4 wq[9], win[3][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 q = *reinterpret_cast<const uint4*>(wt + t * C + co);
            wq[t] = rok[t / 3] ? q : zero4;
        }
        const uint4 bq = *reinterpret_cast<const uint4*>(bias + co);
        // columns w0 - 1, w0, w0 + 1 -> ring slots 3, 0, 1 (column w0 + s sits in slot s & 3)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                win[ky][(kx + 3) & 3] = *reinterpret_cast<const uint4*>(rowp[ky] + (long long)colc(w0 + kx - 1) * C + i * LPP * 8);
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            const int w = w0 + s;
            if (s + 1 < RUN) {          // column w + 2 for the next pixel, into the slot the window left at the last step
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    win[ky][(s + 2) & 3] = *reinterpret_cast<const uint4*>(rowp[ky] + (long long)colc(w + 2) * C + i * LPP * 8);
            }
            __builtin_amdgcn_sched_barrier(0);   // the next column is requested BEFORE this pixel's taps (hipcc would sink the loads to their use)
            const bool lok = w > 0, rgt = w + 1 < W;
            float acc[8], ctr[8];
            un8(bq, acc);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                float xv[8], wv[8];
                un8(win[ky][(s + kx + 3) & 3], xv);
                const uint4 wsel = kx == 0 ? (lok ? wq[t] : zero4) : (kx == 2 ? (rgt ? wq[t] : zero4) : wq[t]);
                un8(wsel, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
                if (t == 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ctr[j] = xv[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = ctr[j] + rbf(acc[j]);
            const uint4 pv = pk8(acc);
            pk[i][s] = pv;
            const uint32_t soff = run_live && w < W ? row_off + (uint32_t)w * (uint32_t)C * 2u : 0xC0000000u;
            __builtin_amdgcn_raw_buffer_store_b128(v4u{pv.x, pv.y, pv.z, pv.w}, rs_y, soff + (uint32_t)(i * LPP * 16), 0, 0);
        }
    }
    // phase 2: LayerNorm of the 8 pixels from the stored bf16 values (the 8 reduction chains are independent: their shuffles overlap)
    uint4 lw[NCH], lb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        lw[i] = *reinterpret_cast<const uint4*>(ln_w + (sub + i * LPP) * 8);
        lb[i] = *reinterpret_cast<const uint4*>(ln_b + (sub + i * LPP) * 8);
    }
#pragma unroll
    for (int s = 0; s < RUN; ++s) {
        const int w = w0 + s;
        float val[NCH][8];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            un8(pk[i][s], val[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sm += val[i][j];
        }
        sm = psum(sm);
        const float mean = sm / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = val[i][j] - mean; q = fmaf(d, d, q); }
        q = psum(q);
        const float rstd = rsqrtf(q / (float)C + eps);
        const uint32_t soff = run_live && w < W ? row_off + (uint32_t)w * (uint32_t)C * 2u : 0xC0000000u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float wf[8], bf[8], o[8];
            un8(lw[i], wf);
            un8(lb[i], bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf((val[i][j] - mean) * rstd, wf[j], bf[j]);
            const uint4 po = pk8(o);
            __builtin_amdgcn_raw_buffer_store_b128(v4u{po.x, po.y, po.z, po.w}, rs_h, soff + (uint32_t)(i * LPP * 16), 0, 0);
        }
    }
}

// col[(oy*Wo+ox), (ky*KW+kx)*C + c] = x[oy*s-p+ky, ox*s-p+kx, c]  (0 outside); row stride ldc >= KH*KW*C
__global__ __launch_bounds__(256) void im2col_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ col, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int ldc, int B,
                                                     const ImgSeg* __restrict__ segs) {
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; Ho = sg.Ho; Wo = sg.Wo; B = 1;
        x += (long long)sg.in_row0 * C; col += (long long)sg.out_row0 * ldc;
    }
    const int chunks = C >> 3;
    const int kk = KH * KW;
    const int HoWo = Ho * Wo;
    const long long total = (long long)B * HoWo * kk * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        long long r = i / chunks;
        const int k = (int)(r % kk);
        const int opix = (int)(r / kk);              // global output pixel over the B images
        const int img = opix / HoWo, lp = opix - img * HoWo;
        const int oy = lp / Wo, ox = lp - oy * Wo;
        const int ky = k / KW, kx = k - ky * KW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        uint4 v = uint4{0, 0, 0, 0};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + ((long long)img * H * W + (long long)iy * W + ix) * C + c * 8);
        *reinterpret_cast<uint4*>(col + (long long)opix * ldc + k * C + c * 8) = v;
    }
}

// xw[(wy*nWx + wx)*ws*ws + iy*ws + ix, c] = x[wy*ws+iy, wx*ws+ix, c]  (0 when outside HxW)
__global__ __launch_bounds__(256) void window_partition_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ xw, int H, int W,
                                                               int C, int ws, int nWy, int nWx, int B, const ImgSeg* __restrict__ segs) {
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];                      // Ho / Wo = window counts, out_row0 = first window row of the image
        H = sg.H; W = sg.W; nWy = sg.Ho; nWx = sg.Wo; B = 1;
        x += (long long)sg.in_row0 * C; xw += (long long)sg.out_row0 * C;
    }
    const int chunks = C >> 3;
    const int nW = nWy * nWx;
    const long long total = (long long)B * nW * ws * ws * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int row = (int)(i / chunks);
        const int gwin = row / (ws * ws), in = row - gwin * ws * ws;
        const int img = gwin / nW, win = gwin - img * nW;
        const int wy = win / nWx, wx = win - wy * nWx;
        const int iy = in / ws, ix = in - iy * ws;
        const int h = wy * ws + iy, w = wx * ws + ix;
        uint4 v = uint4{0, 0, 0, 0};
        if (h < H && w < W) v = *reinterpret_cast<const uint4*>(x + ((long long)img * H * W + (long long)h * W + w) * C + c * 8);
        *reinterpret_cast<uint4*>(xw + (long long)row * C + c * 8) = v;
    }
}

// y[h,w,c] = shortcut[h,w,c] + yw[window row of (h,w), c]
__global__ __launch_bounds__(256) void window_reverse_add_kernel(const uint16_t* __restrict__ yw, const uint16_t* __restrict__ shortcut,
                                                                 uint16_t* __restrict__ y, int H, int W, int C, int ws, int nWx, int nW,
                                                                 int B, const ImgSeg* __restrict__ segs) {
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; nWx = sg.Wo; nW = sg.Ho * sg.Wo; B = 1;
        yw += (long long)sg.out_row0 * C; shortcut += (long long)sg.in_row0 * C; y += (long long)sg.in_row0 * C;
    }
    const int chunks = C >> 3;
    const int HW = H * W;
    const long long total = (long long)B * HW * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int pix = (int)(i / chunks);
        const int img = pix / HW, lp = pix - img * HW;
        const int h = lp / W, w = lp - h * W;
        const int row = (img * nW + (h / ws) * nWx + (w / ws)) * ws * ws + (h % ws) * ws + (w % ws);
        float a[8], b[8];
        un8(*reinterpret_cast<const uint4*>(yw + (long long)row * C + c * 8), a);
        un8(*reinterpret_cast<const uint4*>(shortcut + (long long)pix * C + c * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        *reinterpret_cast<uint4*>(y + (long long)pix * C + c * 8) = pk8(a);
    }
}

// ---- channel attention (group width 32) -------------------------------------------------------
// qkv: [N, 3C] rows = [q | k | v].  Phase 1: partial Gram matrices over token chunks.
//   part[chunk][g][c][c'] = sum_{n in chunk} q[n, g*32+c] * k[n, g*32+c']      (fp32)
constexpr int kCaTok = 256;  // tokens per chunk (round 6: 512 -> 256, four groups per workgroup need the finer grid)
__global__ __launch_bounds__(256) void chattn_gram_kernel(const uint16_t* __restrict__ qkv, int ld, int N, int C, float* __restrict__ part,
                                                          const ImgSeg* __restrict__ segs) {
    __shared__ float sq[64][33];
    __shared__ float sk[64][33];
    const int g = blockIdx.y, chunk = blockIdx.x, G = gridDim.y;
    if (segs) {                                                  // ragged: image z has its own token count (H = N_z) and first row
        N = segs[blockIdx.z].H;
        qkv += (long long)segs[blockIdx.z].in_row0 * ld;
        if (chunk * kCaTok >= N) return;                         // gridDim.x covers the LARGEST image's chunks
    } else {
        qkv += (long long)blockIdx.z * N * ld;                   // image blockIdx.z of the batch: rows [z*N, (z+1)*N)
    }
    part += (long long)blockIdx.z * gridDim.x * G * 1024;
    const int tid = threadIdx.x;
    const int ci = tid >> 4, cj = tid & 15;  // thread owns the 2x2 block (2ci..2ci+1, 2cj..2cj+1)
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    const int n_begin = chunk * kCaTok, n_end = min(N, n_begin + kCaTok);
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
        __syncthreads();
        // 64 tokens x 32 channels of q and k: 64 x 4 chunks each -> 512 16-byte loads, 2 per thread
        for (int t = tid; t < 512; t += 256) {
            const int which = t >> 8, r = (t & 255) >> 2, cc = t & 3;
            float f[8];
            if (n0 + r < n_end) {
                un8(*reinterpret_cast<const uint4*>(qkv + (long long)(n0 + r) * ld + which * C + g * 32 + cc * 8), f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = 0.f;
            }
            float(*dst)[33] = which ? sk : sq;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[r][cc * 8 + j] = f[j];
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const float q0 = sq[r][2 * ci], q1 = sq[r][2 * ci + 1];
            const float k0 = sk[r][2 * cj], k1 = sk[r][2 * cj + 1];
            a00 = fmaf(q0, k0, a00); a01 = fmaf(q0, k1, a01);
            a10 = fmaf(q1, k0, a10); a11 = fmaf(q1, k1, a11);
        }
    }
    float* o = part + (((long long)chunk * G + g) * 32) * 32;
    o[(2 * ci) * 32 + 2 * cj] = a00; o[(2 * ci) * 32 + 2 * cj + 1] = a01;
    o[(2 * ci + 1) * 32 + 2 * cj] = a10; o[(2 * ci + 1) * 32 + 2 * cj + 1] = a11;
}
// Round 6: the same partial Gram matrices on the matrix cores.  The reduction runs over TOKENS, the memory-major index, so the operands
// (8 consecutive tokens of one channel per lane) are read out of a staged [64 tokens][32 channels] bf16 tile column-wise (2-byte LDS reads;
// 72-byte rows put the fragment's two token groups in different bank halves).  A workgroup = FOUR ADJACENT GROUPS of one token chunk, one
// wave per group: together the waves consume whole 128-byte lines of the q and k rows (a group alone reads 64 of a row's 6 KB; with one
// group per workgroup the other half of every line was fetched again by a workgroup on another XCD: PMC fetch 2.1x the footprint).  A wave
// stages and multiplies its own tiles (the next tile's loads in flight meanwhile) and owns its 32 x 32 matrix: no workgroup barrier, no
// cross-wave reduction.  Products of bf16 values are exact in fp32; only the order of the fp32 additions differs from chattn_gram_kernel.
typedef __attribute__((ext_vector_type(8))) __bf16 ca_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ca_f32x16;
constexpr int kCaRow = 72;
__global__ __launch_bounds__(256) void chattn_gram_mfma_kernel(const uint16_t* __restrict__ qkv, int ld, int N, int C, float* __restrict__ part,
                                                               const ImgSeg* __restrict__ segs) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * 64 * kCaRow];      // 36 KB: a q and a k tile per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = C >> 5, g = blockIdx.y * 4 + wave, chunk = blockIdx.x;
    if (segs) {
        N = segs[blockIdx.z].H;
        qkv += (long long)segs[blockIdx.z].in_row0 * ld;
    } else {
        qkv += (long long)blockIdx.z * N * ld;
    }
    if (chunk * kCaTok >= N || g >= G) return;                                    // (no barrier below: a wave may leave alone)
    part += (long long)blockIdx.z * gridDim.x * G * 1024;
    char* sq = smem + wave * (2 * 64 * kCaRow);
    char* sk = sq + 64 * kCaRow;
    const int n_begin = chunk * kCaTok, n_end = min(N, n_begin + kCaTok);
    const int lr = lane >> 2, cc = lane & 3, fi = lane & 31, kg = lane >> 5;
    const uint16_t* base = qkv + g * 32 + cc * 8;
    constexpr int TILES = kCaTok / 64;
    uint4 rq[TILES][4], rk[TILES][4];
    // every tile of the chunk is requested before the first is used (32 loads per lane in flight: the wave's whole share of the chunk);
    // rows past the chunk re-read its last row (a valid address, an L1 hit) and are zeroed when staged
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = min(n_begin + t * 64 + 16 * p + lr, n_end - 1);
            rq[t][p] = *reinterpret_cast<const uint4*>(base + (long long)r * ld);
            rk[t][p] = *reinterpret_cast<const uint4*>(base + (long long)r * ld + C);
        }
    ca_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const int n0 = n_begin + t * 64;
        if (n0 < n_end) {                                                 // wave-uniform
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool ok = n0 + 16 * p + lr < n_end;
                const uint4 a = ok ? rq[t][p] : uint4{0, 0, 0, 0}, b = ok ? rk[t][p] : uint4{0, 0, 0, 0};
                char* dq = sq + (16 * p + lr) * kCaRow + cc * 16;
                char* dk = sk + (16 * p + lr) * kCaRow + cc * 16;
                *reinterpret_cast<uint2*>(dq) = uint2{a.x, a.y}; *reinterpret_cast<uint2*>(dq + 8) = uint2{a.z, a.w};
                *reinterpret_cast<uint2*>(dk) = uint2{b.x, b.y}; *reinterpret_cast<uint2*>(dk + 8) = uint2{b.z, b.w};
            }
            // the tile is wave-private and a wave's LDS instructions execute in order: no fence (a release fence would also wait for the
            // loads of the tiles still in flight), only the compiler is held to the program order
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const char* pq = sq + (16 * m + 8 * kg) * kCaRow + 2 * fi;
                const char* pk = sk + (16 * m + 8 * kg) * kCaRow + 2 * fi;
                uint32_t wa[4], wb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    wa[u] = (uint32_t)*reinterpret_cast<const uint16_t*>(pq + (2 * u) * kCaRow) |
                            ((uint32_t)*reinterpret_cast<const uint16_t*>(pq + (2 * u + 1) * kCaRow) << 16);
                    wb[u] = (uint32_t)*reinterpret_cast<const uint16_t*>(pk + (2 * u) * kCaRow) |
                            ((uint32_t)*reinterpret_cast<const uint16_t*>(pk + (2 * u + 1) * kCaRow) << 16);
                }
                const uint4 ua = uint4{wa[0], wa[1], wa[2], wa[3]}, ub = uint4{wb[0], wb[1], wb[2], wb[3]};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ca_bf16x8, ua), __builtin_bit_cast(ca_bf16x8, ub), acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();                              // the next tile overwrites what other lanes have just read
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* o = part + (((long long)chunk * G + g) * 32) * 32;             // D[q channel][k channel]
#pragma unroll
    for (int r = 0; r < 16; ++r) o[((r >> 2) * 8 + kg * 4 + (r & 3)) * 32 + fi] = acc[r];
}


// Phase 2: A[g][c][:] = softmax_c'( bf16( scale * sum_chunks part ) ) rounded to bf16 (stored fp32).
// One 1024-thread workgroup per group: thread (r, c) sums its element over the chunks in a fixed order
// (4 independent loads in flight), rows are 32-lane halves of a wave.
__global__ __launch_bounds__(1024) void chattn_softmax_kernel(const float* __restrict__ part, int n_chunks, int G, float scale,
                                                              float* __restrict__ A, const ImgSeg* __restrict__ segs) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const int r = tid >> 5, c = tid & 31;
    const long long stride = (long long)G * 1024;
    part += (long long)blockIdx.y * n_chunks * G * 1024;         // image blockIdx.y (n_chunks = the allocation's chunk slots per image)
    if (segs) {                                                  // ragged: this image's own chunk count and q * N^-0.5 (modeling_davit.py:165)
        const int N = segs[blockIdx.y].H;
        n_chunks = (N + kCaTok - 1) / kCaTok;
        scale = 1.0f / sqrtf((float)N);
    }
    A += (long long)blockIdx.y * G * 1024;
    const float* p = part + ((long long)g * 32 + r) * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= n_chunks; k += 4) {
        s0 += p[(long long)k * stride];
        s1 += p[(long long)(k + 1) * stride];
        s2 += p[(long long)(k + 2) * stride];
        s3 += p[(long long)(k + 3) * stride];
    }
    for (; k < n_chunks; ++k) s0 += p[(long long)k * stride];
    float s = rbf(((s0 + s1) + (s2 + s3)) * scale);
    float m = s;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float e = expf(s - m);
    float sum = e;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    A[((long long)g * 32 + r) * 32 + c] = rbf(e / sum);
}

// Phase 3: out[n, g*32 + c] = bf16( sum_c' A[g][c][c'] * v[n, g*32 + c'] )
// One thread = one token of the group: its 32 v values are four 16-byte loads, the 32 x 32 matrix sits in LDS and every lane reads the
// same word of it (a broadcast: no bank conflict), the 32 outputs leave as four 16-byte stores.  Round 3 — the first form staged 8 tokens
// per workgroup pass through LDS between two barriers, with 2-byte global loads and stores: 1.1 TB/s.  Sum order per output unchanged
// (c' ascending, one fmaf chain): bit-identical.
__global__ __launch_bounds__(256) void chattn_apply_kernel(const uint16_t* __restrict__ qkv, int ld, int N, int C, const float* __restrict__ A,
                                                           uint16_t* __restrict__ out, int ldo, const ImgSeg* __restrict__ segs) {
    __shared__ __attribute__((aligned(16))) float sA[32 * 32];
    const int g = blockIdx.y, tid = threadIdx.x;
    if (segs) {
        N = segs[blockIdx.z].H;
        qkv += (long long)segs[blockIdx.z].in_row0 * ld;
        out += (long long)segs[blockIdx.z].in_row0 * ldo;
    } else {
        qkv += (long long)blockIdx.z * N * ld;                   // image blockIdx.z
        out += (long long)blockIdx.z * N * ldo;
    }
    A += (long long)blockIdx.z * gridDim.y * 1024;
    for (int t = tid; t < 1024; t += 256) sA[t] = A[(long long)g * 1024 + t];
    __syncthreads();
    for (int n = blockIdx.x * 256 + tid; n < N; n += gridDim.x * 256) {
        const uint16_t* vp = qkv + (long long)n * ld + 2 * C + g * 32;
        uint4 raw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const uint4*>(vp + j * 8);
        float v[32];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j * 8 + 0] = bf16_lo(raw[j].x); v[j * 8 + 1] = bf16_hi(raw[j].x); v[j * 8 + 2] = bf16_lo(raw[j].y); v[j * 8 + 3] = bf16_hi(raw[j].y);
            v[j * 8 + 4] = bf16_lo(raw[j].z); v[j * 8 + 5] = bf16_hi(raw[j].z); v[j * 8 + 6] = bf16_lo(raw[j].w); v[j * 8 + 7] = bf16_hi(raw[j].w);
        }
        uint16_t* op = out + (long long)n * ldo + g * 32;
        int zoff = 0;       // opaque: the matrix reads are invariant over the token loop and hipcc hoists all 1024 words into registers (256 VGPRs)
        asm volatile("" : "+v"(zoff));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4* row = reinterpret_cast<const float4*>(sA + zoff + (j * 8 + i) * 32);
                float acc = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const float4 a = row[k4];
                    acc = fmaf(a.x, v[k4 * 4 + 0], acc);
                    acc = fmaf(a.y, v[k4 * 4 + 1], acc);
                    acc = fmaf(a.z, v[k4 * 4 + 2], acc);
                    acc = fmaf(a.w, v[k4 * 4 + 3], acc);
                }
                o[i] = acc;
            }
            uint4 w;
            w.x = pack_bf16x2(o[0], o[1]); w.y = pack_bf16x2(o[2], o[3]); w.z = pack_bf16x2(o[4], o[5]); w.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4*>(op + j * 8) = w;
        }
    }
}

// Round 6: out[n, i] = sum_k A[i][k] v[n, k] on the matrix cores — the reduction runs over the group's 32 channels, which are contiguous in
// memory, so the v fragments are plain 16-byte loads (lane = token, 8 channels) and the attention matrix (bf16 values held as fp32) is the
// register-resident other operand.  A lane ends up with 16 of a token's 32 output channels in 4-channel pieces; one exchange with the lane
// 32 away turns them into 16 consecutive channels = two 16-byte stores.  fp32 sums of exact products, rounded to bf16 as chattn_apply_kernel.
__global__ __launch_bounds__(256) void chattn_apply_mfma_kernel(const uint16_t* __restrict__ qkv, int ld, int N, int C, const float* __restrict__ A,
                                                                uint16_t* __restrict__ out, int ldo, const ImgSeg* __restrict__ segs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = C >> 5, g = blockIdx.y * 4 + wave;      // four adjacent groups per workgroup, one per wave: whole 128-byte lines of the v rows
    if (g >= G) return;
    if (segs) {
        N = segs[blockIdx.z].H;
        qkv += (long long)segs[blockIdx.z].in_row0 * ld;
        out += (long long)segs[blockIdx.z].in_row0 * ldo;
    } else {
        qkv += (long long)blockIdx.z * N * ld;
        out += (long long)blockIdx.z * N * ldo;
    }
    A += ((long long)blockIdx.z * G + g) * 1024;
    const int fi = lane & 31, kg = lane >> 5;
    ca_bf16x8 af[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float4 a0 = *reinterpret_cast<const float4*>(A + fi * 32 + 16 * c + 8 * kg), a1 = *reinterpret_cast<const float4*>(A + fi * 32 + 16 * c + 8 * kg + 4);
        const uint4 u = uint4{pack_bf16x2(a0.x, a0.y), pack_bf16x2(a0.z, a0.w), pack_bf16x2(a1.x, a1.y), pack_bf16x2(a1.z, a1.w)};   // exact: bf16 values
        af[c] = __builtin_bit_cast(ca_bf16x8, u);
    }
    // a wave = 4 consecutive 32-token blocks, all 8 loads requested before the first product (one latency per wave instead of four)
    constexpr int NB = 4;
    const int nb0 = blockIdx.x * (32 * NB);
    if (nb0 >= N) return;                       // ragged: the grid covers the largest image
    uint4 v0[NB], v1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = min(nb0 + 32 * b + fi, N - 1);
        const uint16_t* vp = qkv + (long long)n * ld + 2 * C + g * 32 + kg * 8;
        v0[b] = *reinterpret_cast<const uint4*>(vp);
        v1[b] = *reinterpret_cast<const uint4*>(vp + 16);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = nb0 + 32 * b + fi;
        ca_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], __builtin_bit_cast(ca_bf16x8, v0[b]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], __builtin_bit_cast(ca_bf16x8, v1[b]), acc, 0, 0, 0);
        // register r = output channel (r / 4) * 8 + kg * 4 + r % 4 of token n.  The lower half-wave keeps channels 0..15, the upper 16..31:
        // each sends the 8 registers of the other's channels and receives the 4-channel pieces that complete its own 8-channel runs
        // (bit selects, not `kg ? acc[u] : acc[8 + u]`: on the vector type that becomes a run-time element index = a 16-step select chain)
        const uint32_t km = kg ? 0xffffffffu : 0u;
        auto sel = [km](float a, float b) __attribute__((always_inline)) {           // kg ? a : b
            return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, a) & km) | (__builtin_bit_cast(uint32_t, b) & ~km));
        };
        float rcv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rcv[u] = __shfl_xor(sel(acc[u], acc[8 + u]), 32, 64);
        float o[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = sel(rcv[u], acc[u]);                  // channels base + 0..3
            o[4 + u] = sel(acc[8 + u], rcv[u]);          //          base + 4..7
            o[8 + u] = sel(rcv[4 + u], acc[4 + u]);      //          base + 8..11
            o[12 + u] = sel(acc[12 + u], rcv[4 + u]);    //          base + 12..15
        }
        if (n < N) {
            uint16_t* op = out + (long long)n * ldo + g * 32 + kg * 16;
            *reinterpret_cast<uint4*>(op) = uint4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
            *reinterpret_cast<uint4*>(op + 8) = uint4{pack_bf16x2(o[8], o[9]), pack_bf16x2(o[10], o[11]), pack_bf16x2(o[12], o[13]), pack_bf16x2(o[14], o[15])};
        }
    }
}

// dst[(2y+dy)*2W + 2x+dx, co] = src[y*W + x, (dy*2+dx)*Co + co]
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int H, int W, int Co, int B,
                                                             const ImgSeg* __restrict__ segs) {
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; B = 1;
        src += (long long)sg.in_row0 * 4 * Co; dst += (long long)sg.out_row0 * Co;
    }
    const int chunks = Co >> 3;
    const int HW = H * W;
    const long long total = (long long)B * HW * 4 * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        long long r = i / chunks;
        const int q = (int)(r & 3);
        const int pix = (int)(r >> 2);
        const int img = pix / HW, lp = pix - img * HW;
        const int y = lp / W, x = lp - y * W;
        const int dy = q >> 1, dx = q & 1;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (long long)pix * 4 * Co + q * Co + c * 8);
        *reinterpret_cast<uint4*>(dst + ((long long)img * 4 * HW + (long long)(2 * y + dy) * (2 * W) + 2 * x + dx) * Co + c * 8) = v;
    }
}

// y[oy, ox, c] = max over the 2x2 window (floor mode: Ho = H/2, Wo = W/2)
__global__ __launch_bounds__(256) void maxpool2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int H, int W, int C, int B,
                                                       const ImgSeg* __restrict__ segs) {
    if (segs) {
        const ImgSeg sg = segs[blockIdx.y];
        H = sg.H; W = sg.W; B = 1;
        x += (long long)sg.in_row0 * C; y += (long long)sg.out_row0 * C;
    }
    const int Ho = H / 2, Wo = W / 2, chunks = C >> 3;
    const long long total = (long long)B * Ho * Wo * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int pix = (int)(i / chunks);
        const int img = pix / (Ho * Wo), lp = pix - img * (Ho * Wo);
        const int oy = lp / Wo, ox = lp - oy * Wo;
        const long long ib = (long long)img * H * W;
        float m[8], t[8];
        un8(*reinterpret_cast<const uint4*>(x + (ib + (long long)(2 * oy) * W + 2 * ox) * C + c * 8), m);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            un8(*reinterpret_cast<const uint4*>(x + (ib + (long long)(2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C + c * 8), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], t[j]);
        }
        *reinterpret_cast<uint4*>(y + (long long)pix * C + c * 8) = pk8(m);
    }
}

// img: [3, H, W] (bf16 or fp32) -> out [H*W, 8] bf16 (channels 3..7 zero)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_hwc8_kernel(const T* __restrict__ img, uint16_t* __restrict__ out, int H, int W, int B) {
    const long long HW = (long long)H * W, total = HW * B;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const long long b = i / HW, lp = i - b * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (sizeof(T) == 2) f[c] = bf16_to_f32(((const uint16_t*)img)[(b * 3 + c) * HW + lp]);
            else f[c] = ((const float*)img)[(b * 3 + c) * HW + lp];
        }
        *reinterpret_cast<uint4*>(out + i * 8) = pk8(f);
    }
}

// out[r, :] = src_table[kind[r]][index[r], :]   (kind 0: embedding table, 1: image tokens, 2: region tokens)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ t0, const uint16_t* __restrict__ t1,
                                                          const uint16_t* __restrict__ t2, int ld0, int ld1, int ld2,
                                                          const int* __restrict__ plan, uint16_t* __restrict__ out, int ldo, int R, int D) {
    const int chunks = D >> 3;
    const long long total = (long long)R * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int r = (int)(i / chunks);
        const int kind = plan[2 * r], idx = plan[2 * r + 1];
        const uint16_t* src = kind == 0 ? t0 + (long long)idx * ld0 : (kind == 1 ? t1 + (long long)idx * ld1 : t2 + (long long)idx * ld2);
        *reinterpret_cast<uint4*>(out + (long long)r * ldo + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
    }
}

static inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 4096 ? (g > 0 ? g : 1) : 4096);
}

}  // namespace fo1

extern "C" {

int fo1_dwconv3x3_bf16(const void* x, const void* weight9c, const void* bias, void* y, int H, int W, int C, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && weight9c && bias && y && x != y, "dwconv: NULL operand or in-place call");
    FO1_CHECK_ARG(H > 0 && W > 0 && C > 0 && C % 8 == 0 && batch >= 1, "dwconv: bad shape %dx%dx%d x%d", H, W, C, batch);
    FO1_LAUNCH("dwconv3x3", (double)batch * H * W * C * 4.0, dwconv3x3_kernel, dim3(grid_for((long long)batch * H * W * (C / 8))), dim3(256), 0,
               (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)weight9c, (const uint16_t*)bias, (uint16_t*)y, H, W, C, batch);
    return FO1_OK;
}

#ifdef FO1_ENABLE_AB
static int g_dwln_run = 1;      // 0 = per-pixel form everywhere, 1 = product rule, 2 = the run form at every size it exists for (tests)
int fo1_dwconv_ln_set_form(int run_form) { g_dwln_run = run_form < 0 || run_form > 2 ? 1 : run_form; return FO1_OK; }
#else
static constexpr int g_dwln_run = 1;
#endif
// the sliding-window form exists for the widths that fill whole waves (C = 128, 256, 512, 1024): -> chunks, else 0 (per-pixel form)
static int dwln_form(int chunks, long long map_pixels, long long total_pixels) {
    if (!g_dwln_run || map_pixels * chunks * 16 > (1ll << 31)) return 0;       // (the run form's stores address a map of at most 2 GiB)
    if (!(chunks == 16 || chunks == 32 || chunks == 64 || chunks == 128)) return 0;
    // a run is one wave's work for 8 (4 at C = 1024) pixels: below ~16 waves per CU the per-pixel form's 8x more waves fill the chip better
    // (one 48 x 48 x 1024 map: 13 us per-pixel, 21 us in runs; 25 of them: 160 against 112 — profiles/r06_dwconv_run_form_ab.json)
    const long long waves = total_pixels / (chunks == 128 ? 4 : fo1::kDwRun) / (chunks <= 16 ? 4 : (chunks <= 32 ? 2 : 1));
    return waves >= 4096 || g_dwln_run == 2 ? chunks : 0;
}

int fo1_dwconv3x3_ln_bf16(const void* x, const void* weight9c, const void* bias, void* y, const void* ln_weight, const void* ln_bias,
                          float ln_eps, void* h, int H, int W, int C, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && weight9c && bias && y && ln_weight && ln_bias && h && x != y && x != h && y != h,
                  "dwconv_ln: NULL operand or aliased buffers");
    FO1_CHECK_ARG(H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 64 * kDwLnChunks * 8 && batch >= 1, "dwconv_ln: bad shape %dx%dx%d (C <= 2048)", H, W, C);
    const int chunks = C / 8, npix = batch * H * W;
    const uint16_t *xp = (const uint16_t*)x, *wp = (const uint16_t*)weight9c, *bp = (const uint16_t*)bias, *lwp = (const uint16_t*)ln_weight,
                   *lbp = (const uint16_t*)ln_bias;
    const double work = (double)batch * H * W * C * 6.0;
    const long long runs = (long long)batch * H * cdiv(W, chunks == 128 ? 4 : kDwRun);
    FO1_CHECK_ARG(runs < (1ll << 31), "dwconv_ln: too many pixel runs");
#define FO1_DWLN_RUN(PPW, NCH) \
    FO1_LAUNCH("dwconv3x3_ln", work, (dwconv3x3_ln_run_kernel<PPW, NCH, (NCH > 1 ? 4 : 8)>), dim3(cdiv((int)runs, 4 * PPW)), dim3(256), 0, (hipStream_t)stream, xp, wp, bp, \
               (uint16_t*)y, lwp, lbp, (uint16_t*)h, H, W, C, ln_eps, batch, (const ImgSeg*)nullptr)
#define FO1_DWLN_PIX(PPW) \
    FO1_LAUNCH("dwconv3x3_ln", work, dwconv3x3_ln_kernel<PPW>, dim3(cdiv(npix, 4 * PPW)), dim3(256), 0, (hipStream_t)stream, xp, wp, bp, (uint16_t*)y, \
               lwp, lbp, (uint16_t*)h, H, W, C, ln_eps, batch, (const ImgSeg*)nullptr)
    const int form = dwln_form(chunks, npix, npix);
    if (form == 16) { FO1_DWLN_RUN(4, 1); } else if (form == 32) { FO1_DWLN_RUN(2, 1); } else if (form == 64) { FO1_DWLN_RUN(1, 1); }
    else if (form == 128) { FO1_DWLN_RUN(1, 2); }
    else if (chunks <= 16) { FO1_DWLN_PIX(4); } else if (chunks <= 32) { FO1_DWLN_PIX(2); } else { FO1_DWLN_PIX(1); }
#undef FO1_DWLN_RUN
#undef FO1_DWLN_PIX
    return FO1_OK;
}

int fo1_im2col_bf16(const void* x, void* col, int H, int W, int C, int KH, int KW, int stride, int pad, int ld_col, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && col, "im2col: NULL operand");
    FO1_CHECK_ARG(C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "im2col: bad parameters");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    FO1_CHECK_ARG(Ho > 0 && Wo > 0 && ld_col >= KH * KW * C && ld_col % 8 == 0 && batch >= 1, "im2col: bad output shape");
    FO1_LAUNCH("im2col", (double)batch * Ho * Wo * KH * KW * C * 4.0, im2col_kernel, dim3(grid_for((long long)batch * Ho * Wo * KH * KW * (C / 8))),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)col, H, W, C, KH, KW, stride, pad, Ho, Wo, ld_col, batch, (const ImgSeg*)nullptr);
    return FO1_OK;
}

int fo1_window_partition_bf16(const void* x, void* xw, int H, int W, int C, int ws, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && xw && C % 8 == 0 && ws > 0 && batch >= 1, "window_partition: bad arguments");
    const int nWy = cdiv(H, ws), nWx = cdiv(W, ws);
    FO1_LAUNCH("window_partition", (double)batch * nWy * nWx * ws * ws * C * 4.0, window_partition_kernel,
               dim3(grid_for((long long)batch * nWy * nWx * ws * ws * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
               (uint16_t*)xw, H, W, C, ws, nWy, nWx, batch, (const ImgSeg*)nullptr);
    return FO1_OK;
}

int fo1_window_reverse_add_bf16(const void* yw, const void* shortcut, void* y, int H, int W, int C, int ws, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(yw && shortcut && y && C % 8 == 0 && ws > 0 && batch >= 1, "window_reverse: bad arguments");
    FO1_LAUNCH("window_reverse_add", (double)batch * H * W * C * 6.0, window_reverse_add_kernel, dim3(grid_for((long long)batch * H * W * (C / 8))),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)yw, (const uint16_t*)shortcut, (uint16_t*)y, H, W, C, ws, cdiv(W, ws),
               cdiv(H, ws) * cdiv(W, ws), batch, (const ImgSeg*)nullptr);
    return FO1_OK;
}

#ifdef FO1_ENABLE_AB
static int g_chattn_mfma = 1;
int fo1_channel_attention_set_impl(int mfma) { g_chattn_mfma = mfma != 0; return FO1_OK; }
#else
static constexpr int g_chattn_mfma = 1;
#endif

size_t fo1_channel_attention_workspace_bytes(int N, int C, int batch) {
    const int G = C / 32, chunks = fo1::cdiv(N, fo1::kCaTok);
    return (size_t)batch * ((size_t)chunks * G * 1024 + (size_t)G * 1024) * sizeof(float);
}

// qkv rows [batch * N, 3C]: every image's N tokens form their own 32 x 32 per-group attention matrices
int fo1_channel_attention_bf16(const void* qkv, int ld, int N, int C, void* out, int ldo, int batch, void* workspace, size_t workspace_bytes,
                               void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && out && workspace, "channel_attention: NULL operand");
    FO1_CHECK_ARG(N > 0 && C > 0 && C % 32 == 0 && ld >= 3 * C && ld % 8 == 0 && ldo >= C && batch >= 1, "channel_attention: bad shape N=%d C=%d", N, C);
    FO1_CHECK_ARG(ldo % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "channel_attention: rows must be 16-byte aligned (ldo %% 8, qkv / out pointers)");
    if (workspace_bytes < fo1_channel_attention_workspace_bytes(N, C, batch))
        return set_err(FO1_ERR_WORKSPACE, "channel_attention: workspace too small");
    const int G = C / 32, chunks = cdiv(N, kCaTok);
    float* part = (float*)workspace;
    float* A = part + (size_t)batch * chunks * G * 1024;
    hipStream_t st = (hipStream_t)stream;
    if (g_chattn_mfma) {
        FO1_LAUNCH("chattn_gram", (double)batch * N * C * 4.0, chattn_gram_mfma_kernel, dim3(chunks, cdiv(G, 4), batch), dim3(256), 0, st, (const uint16_t*)qkv, ld, N, C, part, (const ImgSeg*)nullptr);
    } else {
        FO1_LAUNCH("chattn_gram", (double)batch * N * C * 4.0, chattn_gram_kernel, dim3(chunks, G, batch), dim3(256), 0, st, (const uint16_t*)qkv, ld, N, C, part, (const ImgSeg*)nullptr);
    }
    // reference: q * N^-0.5 (modeling_davit.py:165)
    FO1_LAUNCH("chattn_softmax", (double)batch * chunks * G * 4096.0, chattn_softmax_kernel, dim3(G, batch), dim3(1024), 0, st, (const float*)part, chunks, G,
               1.0f / sqrtf((float)N), A, (const ImgSeg*)nullptr);
    if (g_chattn_mfma) {
        FO1_LAUNCH("chattn_apply", (double)batch * N * C * 4.0, chattn_apply_mfma_kernel, dim3(cdiv(N, 128), cdiv(G, 4), batch), dim3(256), 0, st, (const uint16_t*)qkv,
                   ld, N, C, (const float*)A, (uint16_t*)out, ldo, (const ImgSeg*)nullptr);
    } else {
        FO1_LAUNCH("chattn_apply", (double)batch * N * C * 4.0, chattn_apply_kernel, dim3(min(cdiv(N, 256), 512), G, batch), dim3(256), 0, st, (const uint16_t*)qkv, ld,
                   N, C, (const float*)A, (uint16_t*)out, ldo, (const ImgSeg*)nullptr);
    }
    return FO1_OK;
}

int fo1_pixel_shuffle2_bf16(const void* src, void* dst, int H, int W, int Co, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(src && dst && Co > 0 && Co % 8 == 0 && batch >= 1, "pixel_shuffle: bad arguments");
    FO1_LAUNCH("pixel_shuffle2", (double)batch * H * W * 4 * Co * 4.0, pixel_shuffle2_kernel, dim3(grid_for((long long)batch * H * W * 4 * (Co / 8))),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, H, W, Co, batch, (const ImgSeg*)nullptr);
    return FO1_OK;
}

int fo1_maxpool2_bf16(const void* x, void* y, int H, int W, int C, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && y && C % 8 == 0 && H >= 2 && W >= 2 && batch >= 1, "maxpool: bad arguments");
    FO1_LAUNCH("maxpool2", (double)batch * H * W * C * 2.5, maxpool2_kernel, dim3(grid_for((long long)batch * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
               (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, H, W, C, batch, (const ImgSeg*)nullptr);
    return FO1_OK;
}

int fo1_nchw_to_hwc8_bf16(const void* img, int is_f32, void* out, int H, int W, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(img && out && H > 0 && W > 0 && batch >= 1, "nchw_to_hwc8: bad arguments");
    if (is_f32)
        FO1_LAUNCH("nchw_to_hwc8", (double)batch * H * W * 28.0, nchw_to_hwc8_kernel<float>, dim3(grid_for((long long)batch * H * W)), dim3(256), 0,
                   (hipStream_t)stream, (const float*)img, (uint16_t*)out, H, W, batch);
    else
        FO1_LAUNCH("nchw_to_hwc8", (double)batch * H * W * 22.0, nchw_to_hwc8_kernel<uint16_t>, dim3(grid_for((long long)batch * H * W)), dim3(256), 0,
                   (hipStream_t)stream, (const uint16_t*)img, (uint16_t*)out, H, W, batch);
    return FO1_OK;
}

int fo1_gather_rows_bf16(const void* table0, int ld0, const void* table1, int ld1, const void* table2, int ld2, const int32_t* plan,
                         void* out, int ldo, int R, int D, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(plan && out && D > 0 && D % 8 == 0 && ldo >= D, "gather_rows: bad arguments");
    if (R == 0) return FO1_OK;
    FO1_LAUNCH("gather_rows", (double)R * D * 4.0, gather_rows_kernel, dim3(grid_for((long long)R * (D / 8))), dim3(256), 0,
               (hipStream_t)stream, (const uint16_t*)table0, (const uint16_t*)table1, (const uint16_t*)table2, ld0, ld1, ld2, plan,
               (uint16_t*)out, ldo, R, D);
    return FO1_OK;
}

// ---- ragged image batches: the same kernels, one workgroup column per image (fo1_img_seg table in device memory) ----------------
static int check_segs(const void* segs, int n_img, const char* what) {
    using namespace fo1;
    FO1_CHECK_ARG(segs && n_img >= 1 && n_img <= 65535, "%s: need a device fo1_img_seg table with 1..65535 images", what);
    return FO1_OK;
}

int fo1_dwconv3x3_ln_var_bf16(const void* x, const void* weight9c, const void* bias, void* y, const void* ln_weight, const void* ln_bias,
                              float ln_eps, void* h, const void* segs, int n_img, int max_pixels, long long total_pixels, int C, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && weight9c && bias && y && ln_weight && ln_bias && h && x != y && x != h && y != h, "dwconv_ln_var: NULL operand or aliased buffers");
    if (int rc = check_segs(segs, n_img, "dwconv_ln_var")) return rc;
    FO1_CHECK_ARG(max_pixels > 0 && C > 0 && C % 8 == 0 && C <= 64 * kDwLnChunks * 8, "dwconv_ln_var: bad shape (C <= 2048)");
    const int chunks = C / 8;
    const ImgSeg* sg = (const ImgSeg*)segs;
    const double work = (double)total_pixels * C * 6.0;
#define FO1_DWLN_VAR(PPW, PIXPB) \
    FO1_LAUNCH("dwconv3x3_ln", work, dwconv3x3_ln_kernel<PPW>, dim3(cdiv(max_pixels, PIXPB), n_img), dim3(256), 0, (hipStream_t)stream, \
               (const uint16_t*)x, (const uint16_t*)weight9c, (const uint16_t*)bias, (uint16_t*)y, (const uint16_t*)ln_weight, \
               (const uint16_t*)ln_bias, (uint16_t*)h, 0, 0, C, ln_eps, 1, sg)
    // the run form's grid: an image has H * ceil(W / 8) <= H * W runs; the workgroups past an image's last run leave at once
#define FO1_DWLN_VAR_RUN(PPW, NCH) \
    FO1_LAUNCH("dwconv3x3_ln", work, (dwconv3x3_ln_run_kernel<PPW, NCH, (NCH > 1 ? 4 : 8)>), dim3(cdiv(max_pixels, 4 * PPW), n_img), dim3(256), 0, (hipStream_t)stream, \
               (const uint16_t*)x, (const uint16_t*)weight9c, (const uint16_t*)bias, (uint16_t*)y, (const uint16_t*)ln_weight, \
               (const uint16_t*)ln_bias, (uint16_t*)h, 0, 0, C, ln_eps, 1, sg)
    const int form = dwln_form(chunks, max_pixels, total_pixels);
    if (form == 16) { FO1_DWLN_VAR_RUN(4, 1); } else if (form == 32) { FO1_DWLN_VAR_RUN(2, 1); } else if (form == 64) { FO1_DWLN_VAR_RUN(1, 1); }
    else if (form == 128) { FO1_DWLN_VAR_RUN(1, 2); }
    else if (chunks <= 16) { FO1_DWLN_VAR(4, 16); } else if (chunks <= 32) { FO1_DWLN_VAR(2, 8); } else { FO1_DWLN_VAR(1, 4); }
#undef FO1_DWLN_VAR_RUN
#undef FO1_DWLN_VAR
    return FO1_OK;
}

int fo1_im2col_var_bf16(const void* x, void* col, const void* segs, int n_img, int max_out_pixels, long long total_out_pixels, int C, int KH, int KW,
                        int stride, int pad, int ld_col, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && col, "im2col_var: NULL operand");
    if (int rc = check_segs(segs, n_img, "im2col_var")) return rc;
    FO1_CHECK_ARG(C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && ld_col >= KH * KW * C && ld_col % 8 == 0 && max_out_pixels > 0,
                  "im2col_var: bad parameters");
    FO1_LAUNCH("im2col", (double)total_out_pixels * KH * KW * C * 4.0, im2col_kernel, dim3(grid_for((long long)max_out_pixels * KH * KW * (C / 8)), n_img),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)col, 0, 0, C, KH, KW, stride, pad, 0, 0, ld_col, 1, (const ImgSeg*)segs);
    return FO1_OK;
}

int fo1_window_partition_var_bf16(const void* x, void* xw, const void* segs, int n_img, int max_window_rows, long long total_window_rows, int C, int ws,
                                  void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && xw && C % 8 == 0 && ws > 0 && max_window_rows > 0, "window_partition_var: bad arguments");
    if (int rc = check_segs(segs, n_img, "window_partition_var")) return rc;
    FO1_LAUNCH("window_partition", (double)total_window_rows * C * 4.0, window_partition_kernel, dim3(grid_for((long long)max_window_rows * (C / 8)), n_img),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)xw, 0, 0, C, ws, 0, 0, 1, (const ImgSeg*)segs);
    return FO1_OK;
}

int fo1_window_reverse_add_var_bf16(const void* yw, const void* shortcut, void* y, const void* segs, int n_img, int max_pixels, long long total_pixels,
                                    int C, int ws, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(yw && shortcut && y && C % 8 == 0 && ws > 0 && max_pixels > 0, "window_reverse_var: bad arguments");
    if (int rc = check_segs(segs, n_img, "window_reverse_var")) return rc;
    FO1_LAUNCH("window_reverse_add", (double)total_pixels * C * 6.0, window_reverse_add_kernel, dim3(grid_for((long long)max_pixels * (C / 8)), n_img),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)yw, (const uint16_t*)shortcut, (uint16_t*)y, 0, 0, C, ws, 0, 0, 1, (const ImgSeg*)segs);
    return FO1_OK;
}

size_t fo1_channel_attention_var_workspace_bytes(int max_tokens, int C, int n_img) { return fo1_channel_attention_workspace_bytes(max_tokens, C, n_img); }

// qkv rows of n_img images of different token counts (segs[i].H = N_i tokens from row segs[i].in_row0): every image its own 32 x 32
// per-group attention matrices and its own q * N_i^-0.5
int fo1_channel_attention_var_bf16(const void* qkv, int ld, const void* segs, int n_img, int max_tokens, long long total_tokens, int C, void* out, int ldo,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(qkv && out && workspace, "channel_attention_var: NULL operand");
    if (int rc = check_segs(segs, n_img, "channel_attention_var")) return rc;
    FO1_CHECK_ARG(max_tokens > 0 && C > 0 && C % 32 == 0 && ld >= 3 * C && ld % 8 == 0 && ldo >= C, "channel_attention_var: bad shape");
    FO1_CHECK_ARG(ldo % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "channel_attention_var: rows must be 16-byte aligned (ldo %% 8, qkv / out pointers)");
    if (workspace_bytes < fo1_channel_attention_workspace_bytes(max_tokens, C, n_img))
        return set_err(FO1_ERR_WORKSPACE, "channel_attention_var: workspace too small");
    const int G = C / 32, chunks = cdiv(max_tokens, kCaTok);
    float* part = (float*)workspace;
    float* A = part + (size_t)n_img * chunks * G * 1024;
    hipStream_t st = (hipStream_t)stream;
    const ImgSeg* sg = (const ImgSeg*)segs;
    if (g_chattn_mfma) {
        FO1_LAUNCH("chattn_gram", (double)total_tokens * C * 4.0, chattn_gram_mfma_kernel, dim3(chunks, cdiv(G, 4), n_img), dim3(256), 0, st, (const uint16_t*)qkv, ld, 0, C, part, sg);
    } else {
        FO1_LAUNCH("chattn_gram", (double)total_tokens * C * 4.0, chattn_gram_kernel, dim3(chunks, G, n_img), dim3(256), 0, st, (const uint16_t*)qkv, ld, 0, C, part, sg);
    }
    FO1_LAUNCH("chattn_softmax", (double)n_img * chunks * G * 4096.0, chattn_softmax_kernel, dim3(G, n_img), dim3(1024), 0, st, (const float*)part, chunks, G,
               0.f, A, sg);
    if (g_chattn_mfma) {
        FO1_LAUNCH("chattn_apply", (double)total_tokens * C * 4.0, chattn_apply_mfma_kernel, dim3(cdiv(max_tokens, 128), cdiv(G, 4), n_img), dim3(256), 0, st,
                   (const uint16_t*)qkv, ld, 0, C, (const float*)A, (uint16_t*)out, ldo, sg);
    } else {
        FO1_LAUNCH("chattn_apply", (double)total_tokens * C * 4.0, chattn_apply_kernel, dim3(min(cdiv(max_tokens, 256), 512), G, n_img), dim3(256), 0, st,
                   (const uint16_t*)qkv, ld, 0, C, (const float*)A, (uint16_t*)out, ldo, sg);
    }
    return FO1_OK;
}

int fo1_pixel_shuffle2_var_bf16(const void* src, void* dst, const void* segs, int n_img, int max_pixels, long long total_pixels, int Co, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(src && dst && Co > 0 && Co % 8 == 0 && max_pixels > 0, "pixel_shuffle_var: bad arguments");
    if (int rc = check_segs(segs, n_img, "pixel_shuffle_var")) return rc;
    FO1_LAUNCH("pixel_shuffle2", (double)total_pixels * 4 * Co * 4.0, pixel_shuffle2_kernel, dim3(grid_for((long long)max_pixels * 4 * (Co / 8)), n_img),
               dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, 0, 0, Co, 1, (const ImgSeg*)segs);
    return FO1_OK;
}

int fo1_maxpool2_var_bf16(const void* x, void* y, const void* segs, int n_img, int max_out_pixels, long long total_out_pixels, int C, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && y && C % 8 == 0 && max_out_pixels > 0, "maxpool_var: bad arguments");
    if (int rc = check_segs(segs, n_img, "maxpool_var")) return rc;
    FO1_LAUNCH("maxpool2", (double)total_out_pixels * 4 * C * 2.5, maxpool2_kernel, dim3(grid_for((long long)max_out_pixels * (C / 8)), n_img), dim3(256), 0,
               (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, 0, 0, C, 1, (const ImgSeg*)segs);
    return FO1_OK;
}

}  // extern "C"
