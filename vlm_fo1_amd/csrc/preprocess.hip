// preprocess.hip — device side of the two image preprocessors.
//
// The host keeps PIL's decode + bicubic resize (bit-exact reproduction of PIL's fixed-point resampler is not attempted);
// everything after it — rescale 1/255, mean/std normalisation, the cast to the tower dtype, and for the primary tower the
// (gh/2, gw/2, 2, 2) merge-block patch order with each patch vector laid out (C=3, T=2, 14, 14) and the single frame
// duplicated along T (reference: Qwen2VLImageProcessor via qwen2_5_vl_encoder.py:206-212; CLIPImageProcessor
// image_processing_clip.py:222-367) — happens here from the uint8 HWC image.  A pixel has 256 possible values per channel, so
// the arithmetic is a 3x256 bf16 look-up table the host builds with the reference's own numpy expression: results are
// bit-identical to "CPU processor, then .to(bfloat16)" by construction, and the upload is 1 byte per sample instead of 4
// (x 2 for the duplicated frame).  HBM-bound byte shuffling: no MFMA, 16-byte stores.
#include "common.h"

namespace fo1 {

// out[row, c*2*P*P + t*P*P + y*P + x] = lut[c][img[(py*P + y), (px*P + x), c]],  row = ((by*gwm + bx)*m + dy)*m + dx,
// py = by*m + dy, px = bx*m + dx.  One thread = 8 consecutive x of one (row, c, y): both T copies written.
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t* __restrict__ img, int W, const uint16_t* __restrict__ lut,
                                                          uint16_t* __restrict__ out, int ld, int gh, int gw, int P, int m) {
    __shared__ uint16_t sl[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) sl[i] = lut[i];
    __syncthreads();
    const int xch = (P + 7) / 8;                       // 8-wide chunks per patch row (14 -> 2: 8 + 6)
    const long long total = (long long)gh * gw * 3 * P * xch;
    const int gwm = gw / m;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xc = (int)(i % xch);
        long long r = i / xch;
        const int y = (int)(r % P); r /= P;
        const int c = (int)(r % 3); r /= 3;
        const int row = (int)r;
        const int dx = row % m, dy = (row / m) % m, blk = row / (m * m);
        const int bx = blk % gwm, by = blk / gwm;
        const int py = by * m + dy, px = bx * m + dx;
        const uint8_t* src = img + ((long long)(py * P + y) * W + px * P + xc * 8) * 3 + c;
        const int n = min(8, P - xc * 8);
        uint16_t* o0 = out + (long long)row * ld + c * 2 * P * P + y * P + xc * 8;
        uint16_t* o1 = o0 + P * P;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n) {
                const uint16_t v = sl[c * 256 + src[j * 3]];
                o0[j] = v;
                o1[j] = v;
            }
    }
}

// out[c, y, x] = lut[c][img[y, x, c]]   (CHW bf16 from HWC uint8); one thread = 8 consecutive x of one (c, y)
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ img, const uint16_t* __restrict__ lut,
                                                           uint16_t* __restrict__ out, int H, int W) {
    __shared__ uint16_t sl[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) sl[i] = lut[i];
    __syncthreads();
    const int xch = (W + 7) / 8;
    const long long total = (long long)3 * H * xch;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xc = (int)(i % xch);
        const long long r = i / xch;
        const int y = (int)(r % H), c = (int)(r / H);
        const uint8_t* src = img + ((long long)y * W + xc * 8) * 3 + c;
        uint16_t* o = out + ((long long)c * H + y) * W + xc * 8;
        const int n = min(8, W - xc * 8);
        for (int j = 0; j < n; ++j) o[j] = sl[c * 256 + src[j * 3]];
    }
}

}  // namespace fo1

extern "C" {

int fo1_patchify_u8_bf16(const void* image_hwc_u8, int H, int W, const void* lut_bf16, void* out, int ld, int patch, int merge,
                         void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(image_hwc_u8 && lut_bf16 && out, "patchify: NULL operand");
    FO1_CHECK_ARG(patch > 0 && merge > 0 && H > 0 && W > 0 && H % (patch * merge) == 0 && W % (patch * merge) == 0,
                  "patchify: %dx%d is not a multiple of patch*merge = %d", H, W, patch * merge);
    FO1_CHECK_ARG(ld >= 3 * 2 * patch * patch, "patchify: row stride %d < %d", ld, 6 * patch * patch);
    const int gh = H / patch, gw = W / patch;
    const long long total = (long long)gh * gw * 3 * patch * ((patch + 7) / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    FO1_LAUNCH("patchify_u8", (double)H * W * 3 * (1 + 4), patchify_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
               (const uint8_t*)image_hwc_u8, W, (const uint16_t*)lut_bf16, (uint16_t*)out, ld, gh, gw, patch, merge);
    return FO1_OK;
}

int fo1_normalize_u8_bf16(const void* image_hwc_u8, int H, int W, const void* lut_bf16, void* out_chw, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(image_hwc_u8 && lut_bf16 && out_chw, "normalize: NULL operand");
    FO1_CHECK_ARG(H > 0 && W > 0, "normalize: empty image");
    const long long total = (long long)3 * H * ((W + 7) / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    FO1_LAUNCH("normalize_u8", (double)H * W * 3 * (1 + 2), normalize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
               (const uint8_t*)image_hwc_u8, (const uint16_t*)lut_bf16, (uint16_t*)out_chw, H, W);
    return FO1_OK;
}

}  // extern "C"
