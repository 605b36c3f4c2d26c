// swin_ops.hip — data-movement and normalisation kernels of the UPN detector's Swin-L backbone and input projections (SURVEY 8f
// rank 4) for gfx950.  Reference: detect_tools/upn/models/backbone/swin.py — SwinTransformerBlock.forward :259-318 (pad to the
// window multiple AFTER norm1, cyclic shift by torch.roll, window_partition :42-55 / window_reverse :58-74), PatchMerging :333-357
// — and models/architecture/upn_model.py:246-262 (input_proj = Conv2d + GroupNorm(32, 256)).
// Token-major bf16 rows [H*W, C] throughout; 16-byte vectors; all HBM-bound copies (bytes moved = the work reported).
#include "common.h"

namespace fo1 {

__device__ __forceinline__ void sw_un8(const uint4& v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 sw_pk8(const float (&f)[8]) {
    return uint4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}

// xw[window (wy, wx), local (iy, ix)] = padded_x[(wy*ws + iy + shift) mod Hp][(wx*ws + ix + shift) mod Wp]   (zero outside H x W)
__global__ __launch_bounds__(256) void swin_partition_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ xw, int H, int W, int C, int ws,
                                                             int shift, int nWy, int nWx, int B) {
    const int chunks = C >> 3, nW = nWy * nWx, Hp = nWy * ws, Wp = nWx * ws;
    const long long total = (long long)B * nW * ws * ws * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int row = (int)(i / chunks);
        const int gwin = row / (ws * ws), in = row - gwin * ws * ws;
        const int img = gwin / nW, win = gwin - img * nW;
        const int wy = win / nWx, wx = win - wy * nWx;
        const int iy = in / ws, ix = in - iy * ws;
        int h = wy * ws + iy + shift, w = wx * ws + ix + shift;
        h = h >= Hp ? h - Hp : h;
        w = w >= Wp ? w - Wp : w;
        uint4 v = uint4{0, 0, 0, 0};
        if (h < H && w < W) v = *reinterpret_cast<const uint4*>(x + ((long long)img * H * W + (long long)h * W + w) * C + c * 8);
        *reinterpret_cast<uint4*>(xw + (long long)row * C + c * 8) = v;
    }
}

// y[h, w] = shortcut[h, w] + yw[row of ((h - shift) mod Hp, (w - shift) mod Wp)]
__global__ __launch_bounds__(256) void swin_reverse_add_kernel(const uint16_t* __restrict__ yw, const uint16_t* __restrict__ shortcut, uint16_t* __restrict__ y,
                                                               int H, int W, int C, int ws, int shift, int nWy, int nWx, int B) {
    const int chunks = C >> 3, HW = H * W, nW = nWy * nWx, Hp = nWy * ws, Wp = nWx * ws;
    const long long total = (long long)B * HW * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int pix = (int)(i / chunks);
        const int img = pix / HW, lp = pix - img * HW;
        const int h = lp / W, w = lp - h * W;
        int hs = h - shift, wsft = w - shift;
        hs = hs < 0 ? hs + Hp : hs;
        wsft = wsft < 0 ? wsft + Wp : wsft;
        const int row = (img * nW + (hs / ws) * nWx + (wsft / ws)) * ws * ws + (hs % ws) * ws + (wsft % ws);
        float a[8], b[8];
        sw_un8(*reinterpret_cast<const uint4*>(yw + (long long)row * C + c * 8), a);
        sw_un8(*reinterpret_cast<const uint4*>(shortcut + (long long)pix * C + c * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        *reinterpret_cast<uint4*>(y + (long long)pix * C + c * 8) = sw_pk8(a);
    }
}

// PatchMerging gather: out[(i, j)] = [x(2i, 2j) | x(2i+1, 2j) | x(2i, 2j+1) | x(2i+1, 2j+1)]  (zero beyond an odd edge)
__global__ __launch_bounds__(256) void patch_merge_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int H, int W, int C, int B) {
    const int chunks = C >> 3, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * Ho * Wo * 4 * chunks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        long long r = i / chunks;
        const int q = (int)(r & 3);
        r >>= 2;
        const int img = (int)(r / (Ho * Wo)), lp = (int)(r - (long long)img * Ho * Wo);
        const int oi = lp / Wo, oj = lp - oi * Wo;
        const int h = 2 * oi + (q & 1), w = 2 * oj + (q >> 1);
        uint4 v = uint4{0, 0, 0, 0};
        if (h < H && w < W) v = *reinterpret_cast<const uint4*>(x + ((long long)img * H * W + (long long)h * W + w) * C + c * 8);
        *reinterpret_cast<uint4*>(out + (r * 4 + q) * C + c * 8) = v;
    }
}

// ---- GroupNorm over a token-major map: statistics per (image, group) over all S tokens x C/G channels ------------------------
// pass 1: partial (sum, sum of squares) per (token chunk, group) in fp32;  pass 2: every workgroup folds the partials of its
// group in chunk order (fixed order: deterministic) and normalises.  Biased variance, eps inside the sqrt (nn.GroupNorm).
constexpr int kGnTok = 256;   // tokens per chunk
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const uint16_t* __restrict__ x, int ld, int S, int C, int G, float* __restrict__ part) {
    __shared__ float s_a[4], s_b[4];
    const int chunk = blockIdx.x, g = blockIdx.y, cg = C / G;
    const int t0 = chunk * kGnTok, t1 = min(S, t0 + kGnTok);
    float a = 0.f, b = 0.f;
    const int per = cg >> 3;                                         // 16-byte pieces per token in this group
    for (int i = threadIdx.x; i < (t1 - t0) * per; i += 256) {
        const int t = t0 + i / per, pc = i - (i / per) * per;
        float f[8];
        sw_un8(*reinterpret_cast<const uint4*>(x + (long long)t * ld + g * cg + pc * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a += f[j]; b += f[j] * f[j]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long long)chunk * G + g) * 2] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        part[((long long)chunk * G + g) * 2 + 1] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const uint16_t* __restrict__ x, int ld, int S, int C, int G, const float* __restrict__ part,
                                                              int n_chunks, const uint16_t* __restrict__ weight, const uint16_t* __restrict__ bias,
                                                              float eps, uint16_t* __restrict__ y, int ldy) {
    __shared__ float s_mean, s_rstd;
    const int chunk = blockIdx.x, g = blockIdx.y, cg = C / G;
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < n_chunks; ++k) { a += part[((long long)k * G + g) * 2]; b += part[((long long)k * G + g) * 2 + 1]; }
        const float n = (float)S * (float)cg;
        const float mean = a / n;
        const float var = fmaxf(b / n - mean * mean, 0.f);
        s_mean = mean;
        s_rstd = rsqrtf(var + eps);
    }
    __syncthreads();
    const float mean = s_mean, rstd = s_rstd;
    const int t0 = chunk * kGnTok, t1 = min(S, t0 + kGnTok);
    const int per = cg >> 3;
    for (int i = threadIdx.x; i < (t1 - t0) * per; i += 256) {
        const int t = t0 + i / per, pc = i - (i / per) * per;
        float f[8], w[8], b[8];
        sw_un8(*reinterpret_cast<const uint4*>(x + (long long)t * ld + g * cg + pc * 8), f);
        sw_un8(*reinterpret_cast<const uint4*>(weight + g * cg + pc * 8), w);
        sw_un8(*reinterpret_cast<const uint4*>(bias + g * cg + pc * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * w[j] + b[j];
        *reinterpret_cast<uint4*>(y + (long long)t * ldy + g * cg + pc * 8) = sw_pk8(f);
    }
}

static inline int sw_grid(long long n) { const long long g = (n + 255) / 256; return (int)(g < 8192 ? g : 8192); }

}  // namespace fo1

extern "C" {

int fo1_swin_window_partition_bf16(const void* x, void* xw, int H, int W, int C, int ws, int shift, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && xw && H > 0 && W > 0 && C % 8 == 0 && ws > 0 && shift >= 0 && shift < ws && batch >= 1, "swin_window_partition: bad arguments");
    const int nWy = cdiv(H, ws), nWx = cdiv(W, ws);
    const long long n = (long long)batch * nWy * nWx * ws * ws * (C / 8);
    FO1_LAUNCH("swin_partition", (double)n * 32.0, swin_partition_kernel, dim3(sw_grid(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)xw, H,
               W, C, ws, shift, nWy, nWx, batch);
    return FO1_OK;
}

int fo1_swin_window_reverse_add_bf16(const void* yw, const void* shortcut, void* y, int H, int W, int C, int ws, int shift, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(yw && shortcut && y && H > 0 && W > 0 && C % 8 == 0 && ws > 0 && shift >= 0 && shift < ws && batch >= 1, "swin_window_reverse: bad arguments");
    const long long n = (long long)batch * H * W * (C / 8);
    FO1_LAUNCH("swin_reverse_add", (double)n * 48.0, swin_reverse_add_kernel, dim3(sw_grid(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)yw,
               (const uint16_t*)shortcut, (uint16_t*)y, H, W, C, ws, shift, cdiv(H, ws), cdiv(W, ws), batch);
    return FO1_OK;
}

int fo1_patch_merge_bf16(const void* x, void* out, int H, int W, int C, int batch, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && out && H > 0 && W > 0 && C % 8 == 0 && batch >= 1, "patch_merge: bad arguments");
    const long long n = (long long)batch * ((H + 1) / 2) * ((W + 1) / 2) * 4 * (C / 8);
    FO1_LAUNCH("patch_merge", (double)n * 32.0, patch_merge_kernel, dim3(sw_grid(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)out, H, W, C,
               batch);
    return FO1_OK;
}

size_t fo1_groupnorm_tokens_workspace_bytes(int S, int groups) { return (size_t)fo1::cdiv(S, fo1::kGnTok) * groups * 2 * sizeof(float); }

// nn.GroupNorm(groups, C) over one image's token-major map x [S, C] (statistics over all S tokens x C/groups channels of a group)
int fo1_groupnorm_tokens_bf16(const void* x, int ldx, int S, int C, int groups, const void* weight, const void* bias, float eps, void* y, int ldy,
                              void* workspace, size_t workspace_bytes, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(x && y && weight && bias && workspace && S > 0 && C > 0 && groups > 0 && C % groups == 0 && (C / groups) % 8 == 0 && ldx >= C && ldy >= C &&
                      ldx % 8 == 0 && ldy % 8 == 0,
                  "groupnorm_tokens: bad arguments (S=%d C=%d groups=%d)", S, C, groups);
    if (workspace_bytes < fo1_groupnorm_tokens_workspace_bytes(S, groups)) return set_err(FO1_ERR_WORKSPACE, "groupnorm_tokens: workspace too small");
    const int n_chunks = cdiv(S, kGnTok);
    hipStream_t st = (hipStream_t)stream;
    FO1_LAUNCH("groupnorm_partial", (double)S * C * 2.0, groupnorm_partial_kernel, dim3(n_chunks, groups), dim3(256), 0, st, (const uint16_t*)x, ldx, S, C, groups,
               (float*)workspace);
    FO1_LAUNCH("groupnorm_apply", (double)S * C * 4.0, groupnorm_apply_kernel, dim3(n_chunks, groups), dim3(256), 0, st, (const uint16_t*)x, ldx, S, C, groups,
               (const float*)workspace, n_chunks, (const uint16_t*)weight, (const uint16_t*)bias, eps, (uint16_t*)y, ldy);
    return FO1_OK;
}

}  // extern "C"
