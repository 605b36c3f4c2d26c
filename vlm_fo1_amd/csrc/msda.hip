// msda.hip — multi-scale deformable attention forward for gfx950 (the reference's only native operator,
// detect_tools/upn/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 + the bilinear helper :32-84; host shapes
// ms_deform_attn_cuda.cu:25-80).  For query q, head m:
//     out[n, q, m, :] = sum_{level l, point p} weight[n,q,m,l,p] * bilinear(value[n, level l, :, m, :], loc[n,q,m,l,p])
// bilinear = 4 taps around (loc_y * H - 0.5, loc_x * W - 0.5), taps outside the map read zero, samples at or beyond one
// cell outside the map are skipped (align_corners = False grid_sample with zero padding).
//
// A gather over an L2 / Infinity-Cache resident value tensor (UPN: 25 MB) — no MFMA.  A thread owns VEC consecutive channels of one
// (n, q, m): every tap is one 16-byte load (fp32 x 4, bf16 x 8), a head's 32 channels are one contiguous 128 / 64-byte row piece,
// consecutive lanes cover consecutive channels then consecutive heads (value is [N, S, M, D]: a query's M heads at one pixel are
// contiguous too).  The L*P (x, y, weight) triples of a (query, head) are read by every lane of its group from the same addresses
// (one broadcast cache line).  fp32 accumulation in the reference's operation order (w1 v1 + w2 v2 + w3 v3 + w4 v4, then * weight).
#include "common.h"

namespace fo1 {

template <typename T> struct MsdaAcc { typedef float type; };
template <> struct MsdaAcc<double> { typedef double type; };

template <typename VT, int VEC> struct MsdaVec;
template <> struct MsdaVec<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]}; }
};
template <> struct MsdaVec<float, 1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <> struct MsdaVec<double, 1> {
    static __device__ __forceinline__ void load(const double* p, double (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(double* p, const double (&v)[1]) { *p = v[0]; }
};
template <> struct MsdaVec<uint16_t, 8> {   // bf16
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = bf16_lo(t.x); v[1] = bf16_hi(t.x); v[2] = bf16_lo(t.y); v[3] = bf16_hi(t.y);
        v[4] = bf16_lo(t.z); v[5] = bf16_hi(t.z); v[6] = bf16_lo(t.w); v[7] = bf16_hi(t.w);
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
};
template <> struct MsdaVec<uint16_t, 1> {
    static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[1]) { v[0] = bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }
};

// VT: value / out element (float, double, uint16_t = bf16); LT: loc / weight element (float or double)
template <typename VT, typename LT, int VEC>
__global__ __launch_bounds__(256) void msda_forward_kernel(const VT* __restrict__ value, const long long* __restrict__ shapes,
                                                           const long long* __restrict__ level_start, const LT* __restrict__ loc,
                                                           const LT* __restrict__ weight, int S, int M, int D, int L, int Lq, int P,
                                                           long long total, VT* __restrict__ out) {
    typedef typename MsdaAcc<LT>::type AT;
    const int tpi = D / VEC;                                   // threads per (n, q, m)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long item = e / tpi;                            // (n * Lq + q) * M + m
    const int c0 = (int)(e - item * tpi) * VEC;
    const int m = (int)(item % M);
    const long long n = item / ((long long)M * Lq);
    const LT* lp = loc + item * L * P * 2;
    const LT* wp = weight + item * L * P;
    const long long w_stride = (long long)M * D;
    AT col[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const VT* vb = value + (n * S + level_start[l]) * w_stride + (long long)m * D + c0;
        const long long h_stride = (long long)W * w_stride;
        for (int p = 0; p < P; ++p) {
            const AT loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1], aw = wp[l * P + p];
            const AT h_im = loc_h * H - (AT)0.5, w_im = loc_w * W - (AT)0.5;
            // Branch-free: all four taps are always loaded from clamped (valid) addresses and the weights of taps / samples outside
            // the map are zeroed — a load under a per-lane condition costs a serialised round trip each with hipcc.
            const bool inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const AT hf = floor(h_im), wf = floor(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const int h_high = h_low + 1, w_high = w_low + 1;
            const AT lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
            const bool hl_ok = h_low >= 0 && h_low <= H - 1, hh_ok = h_high >= 0 && h_high <= H - 1;
            const bool wl_ok = w_low >= 0 && w_low <= W - 1, wh_ok = w_high >= 0 && w_high <= W - 1;
            const int hl = min(max(h_low, 0), H - 1), hh_i = min(max(h_high, 0), H - 1);
            const int wl = min(max(w_low, 0), W - 1), wh_i = min(max(w_high, 0), W - 1);
            AT v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            MsdaVec<VT, VEC>::load(vb + hl * h_stride + wl * w_stride, v1);
            MsdaVec<VT, VEC>::load(vb + hl * h_stride + wh_i * w_stride, v2);
            MsdaVec<VT, VEC>::load(vb + hh_i * h_stride + wl * w_stride, v3);
            MsdaVec<VT, VEC>::load(vb + hh_i * h_stride + wh_i * w_stride, v4);
            const AT w1 = (hl_ok && wl_ok) ? hh * hw : (AT)0, w2 = (hl_ok && wh_ok) ? hh * lw : (AT)0;
            const AT w3 = (hh_ok && wl_ok) ? lh * hw : (AT)0, w4 = (hh_ok && wh_ok) ? lh * lw : (AT)0;
            const AT aws = inside ? aw : (AT)0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) col[i] += (w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i]) * aws;
        }
    }
    MsdaVec<VT, VEC>::store(out + item * D + c0, col);
}

// ---- fused form: softmax + sampling locations + gather in one launch ------------------------------------------------------------
// MSDeformAttn.forward (ops/modules/ms_deform_attn.py:100-204) materialises sampling_offsets -> sampling_locations [N,Lq,M,L,P,2] and
// softmax(attention_weights) [N,Lq,M,L,P] between its Linear layers and the operator (42 MB per UPN encoder layer).  Here the
// operator takes the RAW output of the two Linear layers (one GEMM, fp32: [offsets M*L*P*2 | logits M*L*P] per query) and the
// reference points, and forms on the fly
//     loc = ref[l] + off / (W_l, H_l)                         (2-d reference points, :150-157)
//     loc = ref[l][:2] + off / P * ref[l][2:] * 0.5           (4-d reference boxes, :169-175)
//     w   = softmax over the head's L*P logits                (:143-147)
// value / out are bf16 (engine form), everything else fp32.  A thread owns 8 channels of one (query, head); the softmax statistics
// are recomputed by the D/8 threads of the head (L*P exps, from one cached line).
template <int RD>
__global__ __launch_bounds__(256) void msda_fused_kernel(const uint16_t* __restrict__ value, const long long* __restrict__ shapes,
                                                         const long long* __restrict__ level_start, const float* __restrict__ ol,
                                                         const float* __restrict__ ref, int ref_levels, int S, int M, int D, int L, int Lq,
                                                         int P, long long total, uint16_t* __restrict__ out) {
    constexpr int VEC = 8;
    const int tpi = D / VEC;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long item = e / tpi;                            // (n * Lq + q) * M + m
    const int c0 = (int)(e - item * tpi) * VEC;
    const int m = (int)(item % M);
    const long long nq = item / M;                             // n * Lq + q
    const long long n = nq / Lq;
    const int LP = L * P;
    const float* row = ol + nq * ((long long)M * LP * 3);
    const float* offp = row + (long long)m * LP * 2;           // [L][P][2]
    const float* lgp = row + (long long)M * LP * 2 + (long long)m * LP;
    const float* rp = ref + nq * ((long long)ref_levels * RD);      // ref_levels == 1: one point / box for every level
    float mx = -INFINITY;
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lgp[i]);
    float den = 0.f;
    for (int i = 0; i < LP; ++i) den += expf(lgp[i] - mx);
    const float inv_den = 1.0f / den;
    const long long w_stride = (long long)M * D;
    float col[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) col[i] = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const uint16_t* vb = value + (n * S + level_start[l]) * w_stride + (long long)m * D + c0;
        const long long h_stride = (long long)W * w_stride;
        const int rl = ref_levels == 1 ? 0 : l;
        const float rx = rp[rl * RD], ry = rp[rl * RD + 1];
        const float rw = RD == 4 ? rp[rl * RD + 2] : 0.f, rh = RD == 4 ? rp[rl * RD + 3] : 0.f;
        for (int p = 0; p < P; ++p) {
            const float ox = offp[(l * P + p) * 2], oy = offp[(l * P + p) * 2 + 1];
            const float aw = expf(lgp[l * P + p] - mx) * inv_den;
            float loc_w, loc_h;
            if (RD == 2) { loc_w = rx + ox / (float)W; loc_h = ry + oy / (float)H; }
            else { loc_w = rx + ox / (float)P * rw * 0.5f; loc_h = ry + oy / (float)P * rh * 0.5f; }
            const float h_im = loc_h * H - 0.5f, w_im = loc_w * W - 0.5f;
            const bool inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
            const bool hl_ok = h_low >= 0 && h_low <= H - 1, hh_ok = h_high >= 0 && h_high <= H - 1;
            const bool wl_ok = w_low >= 0 && w_low <= W - 1, wh_ok = w_high >= 0 && w_high <= W - 1;
            const int hl = min(max(h_low, 0), H - 1), hh_i = min(max(h_high, 0), H - 1);
            const int wl = min(max(w_low, 0), W - 1), wh_i = min(max(w_high, 0), W - 1);
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            MsdaVec<uint16_t, VEC>::load(vb + hl * h_stride + wl * w_stride, v1);
            MsdaVec<uint16_t, VEC>::load(vb + hl * h_stride + wh_i * w_stride, v2);
            MsdaVec<uint16_t, VEC>::load(vb + hh_i * h_stride + wl * w_stride, v3);
            MsdaVec<uint16_t, VEC>::load(vb + hh_i * h_stride + wh_i * w_stride, v4);
            const float w1 = (hl_ok && wl_ok) ? hh * hw : 0.f, w2 = (hl_ok && wh_ok) ? hh * lw : 0.f;
            const float w3 = (hh_ok && wl_ok) ? lh * hw : 0.f, w4 = (hh_ok && wh_ok) ? lh * lw : 0.f;
            const float aws = inside ? aw : 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) col[i] += (w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i]) * aws;
        }
    }
    MsdaVec<uint16_t, VEC>::store(out + item * D + c0, col);
}

template <typename VT, typename LT, int VEC>
static int launch_msda(const void* value, const long long* shapes, const long long* start, const void* loc, const void* weight, int N, int S,
                       int M, int D, int L, int Lq, int P, void* out, hipStream_t st) {
    const long long total = (long long)N * Lq * M * (D / VEC);
    const long long grid = (total + 255) / 256;
    // bytes gathered: 4 taps x D channels per (query, head, level, point) + the output (the figure a roofline is quoted on)
    const double work = (double)N * Lq * M * ((double)L * P * (4.0 * D * sizeof(VT) + 3.0 * sizeof(LT)) + (double)D * sizeof(VT));
    FO1_LAUNCH("msda_forward", work, (msda_forward_kernel<VT, LT, VEC>), dim3((unsigned)grid), dim3(256), 0, st, (const VT*)value, shapes, start,
               (const LT*)loc, (const LT*)weight, S, M, D, L, Lq, P, total, (VT*)out);
    return FO1_OK;
}

}  // namespace fo1

extern "C" {

// Multi-scale deformable attention forward (the reference's MSDA.ms_deform_attn_forward, ops/src/ms_deform_attn.h:21-36; the
// im2col_step argument only batches the reference's launches and is not needed).
//   value [N, S, M, D]; spatial_shapes int64 [L, 2] = (H_l, W_l) and level_start_index int64 [L] ON THE DEVICE (as the reference
//   passes them); sampling_loc [N, Lq, M, L, P, 2] = (x, y) in [0, 1]; attn_weight [N, Lq, M, L, P]; out [N, Lq, M * D].
//   dtype 0: everything fp32 (the reference's default).  1: everything fp64 (the reference test's double check).
//   2: value / out bf16, sampling_loc / attn_weight fp32 (engine form), fp32 accumulation.
int fo1_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* sampling_loc,
                               const void* attn_weight, int N, int S, int M, int D, int L, int Lq, int P, void* out, int dtype, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, "ms_deform_attn: NULL operand");
    FO1_CHECK_ARG(N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && L <= 64 && Lq > 0 && P > 0, "ms_deform_attn: bad shape N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", N,
                  S, M, D, L, Lq, P);
    FO1_CHECK_ARG(dtype >= 0 && dtype <= 2, "ms_deform_attn: dtype %d (0 fp32, 1 fp64, 2 bf16 value)", dtype);
    hipStream_t st = (hipStream_t)stream;
    const long long* sh = (const long long*)spatial_shapes;
    const long long* ls = (const long long*)level_start_index;
    const bool a16 = (((uintptr_t)value | (uintptr_t)out) & 15) == 0;
    if (dtype == 0) {
        if (D % 4 == 0 && a16) return launch_msda<float, float, 4>(value, sh, ls, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
        return launch_msda<float, float, 1>(value, sh, ls, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
    }
    if (dtype == 1) return launch_msda<double, double, 1>(value, sh, ls, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
    if (D % 8 == 0 && a16) return launch_msda<uint16_t, float, 8>(value, sh, ls, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
    return launch_msda<uint16_t, float, 1>(value, sh, ls, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
}

// Fused MSDeformAttn core (engine form): raw [offsets | logits] rows of the module's two Linear layers + reference points ->
// attended rows, bf16 values.  offsets_logits fp32 [N, Lq, M*L*P*3] (first M*L*P*2 = sampling_offsets(query) viewed [M][L][P][2],
// then M*L*P = attention_weights(query) viewed [M][L*P], ops/modules/ms_deform_attn.py:135-147); reference_points fp32
// [N, Lq, ref_levels, ref_dim], ref_levels = L or 1 (the same point / box at every level: an unpadded image's valid ratios are 1),
// ref_dim 2 (points) or 4 (cx, cy, w, h boxes; the default normaliser of :169-175); value bf16 [N, S, M*D]; out bf16 [N, Lq, M*D].
int fo1_msda_fused_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* offsets_logits,
                        const float* reference_points, int ref_levels, int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, void* out,
                        void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(value && spatial_shapes && level_start_index && offsets_logits && reference_points && out, "msda_fused: NULL operand");
    FO1_CHECK_ARG(N > 0 && S > 0 && M > 0 && D > 0 && D % 8 == 0 && L > 0 && L <= 64 && Lq > 0 && P > 0, "msda_fused: bad shape N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d",
                  N, S, M, D, L, Lq, P);
    FO1_CHECK_ARG((ref_dim == 2 || ref_dim == 4) && (ref_levels == 1 || ref_levels == L), "msda_fused: reference points must be 2-d or 4-d, for 1 or L levels (got %d, %d)", ref_dim, ref_levels);
    FO1_CHECK_ARG((((uintptr_t)value | (uintptr_t)out) & 15) == 0, "msda_fused: value / out must be 16-byte aligned");
    const long long total = (long long)N * Lq * M * (D / 8);
    const long long grid = (total + 255) / 256;
    const double work = (double)N * Lq * M * ((double)L * P * (4.0 * D * 2 + 12.0) + (double)D * 2);
    hipStream_t st = (hipStream_t)stream;
    if (ref_dim == 2)
        FO1_LAUNCH("msda_fused", work, msda_fused_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, (const uint16_t*)value, (const long long*)spatial_shapes,
                   (const long long*)level_start_index, offsets_logits, reference_points, ref_levels, S, M, D, L, Lq, P, total, (uint16_t*)out);
    else
        FO1_LAUNCH("msda_fused", work, msda_fused_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, (const uint16_t*)value, (const long long*)spatial_shapes,
                   (const long long*)level_start_index, offsets_logits, reference_points, ref_levels, S, M, D, L, Lq, P, total, (uint16_t*)out);
    return FO1_OK;
}

}  // extern "C"
