/*
 * ORACLE — test infrastructure only.  Never imported by the product path (vlm_fo1_amd/, vlm_fo1/, detect_tools/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * CPU restatement of the forward pass of the reference's only native operator: multi-scale deformable attention
 * (UPN proposal detector).  Follows
 *   detect_tools/upn/ops/src/cuda/ms_deform_im2col_cuda.cuh:32-84   ms_deform_attn_im2col_bilinear (zero outside, 4 taps)
 *   detect_tools/upn/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 ms_deformable_im2col_gpu_kernel (one output element =
 *       sum over levels and points of weight * bilinear(value_level, loc * (W, H) - 0.5), samples with
 *       h_im <= -1 || w_im <= -1 || h_im >= H || w_im >= W skipped)
 *   detect_tools/upn/ops/src/cuda/ms_deform_attn_cuda.cu:25-80      tensor shapes / im2col_step batching (a no-op for the result)
 * in the same operation order (w1 v1 + w2 v2 + w3 v3 + w4 v4, then * weight, accumulated level-major, point-minor).
 * The reference's CPU entry (ops/src/cpu/ms_deform_attn_cpu.cpp) is an AT_ERROR stub, so there is no C/C++ reference to compile;
 * the restatement is pinned against the reference's own pure-torch ms_deform_attn_core_pytorch
 * (ops/functions/ms_deform_attn_func.py:41-61, what ops/test.py compares the CUDA op with) imported in place:
 * tests/golden/make_msda_golden.py -> tests/golden/msda_ref.npz, checked by tests/test_oracle_msda.py.
 *
 * value [N][S][M][D], spatial_shapes int64 [L][2] = (H, W), level_start int64 [L], loc [N][Lq][M][L][P][2] = (x, y) in [0, 1],
 * weight [N][Lq][M][L][P], out [N][Lq][M*D].
 */
#include <math.h>
#include <stdint.h>

#define MSDA_IMPL(NAME, T, FLOOR)                                                                                              \
    static T NAME##_bilinear(const T* v, int H, int W, int M, int D, T h, T w, int m, int c) {                                 \
        const int h_low = (int)FLOOR(h), w_low = (int)FLOOR(w);                                                                \
        const int h_high = h_low + 1, w_high = w_low + 1;                                                                      \
        const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                                      \
        const long long w_stride = (long long)M * D, h_stride = (long long)W * w_stride, base = (long long)m * D + c;          \
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                                      \
        if (h_low >= 0 && w_low >= 0) v1 = v[h_low * h_stride + w_low * w_stride + base];                                      \
        if (h_low >= 0 && w_high <= W - 1) v2 = v[h_low * h_stride + w_high * w_stride + base];                                \
        if (h_high <= H - 1 && w_low >= 0) v3 = v[h_high * h_stride + w_low * w_stride + base];                                \
        if (h_high <= H - 1 && w_high <= W - 1) v4 = v[h_high * h_stride + w_high * w_stride + base];                          \
        const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                                        \
        return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                                                        \
    }                                                                                                                          \
    void NAME(const T* value, const int64_t* shapes, const int64_t* level_start, const T* loc, const T* weight, int N, int S,  \
              int M, int D, int L, int Lq, int P, T* out) {                                                                    \
        for (int n = 0; n < N; ++n)                                                                                            \
            for (int q = 0; q < Lq; ++q)                                                                                       \
                for (int m = 0; m < M; ++m)                                                                                    \
                    for (int c = 0; c < D; ++c) {                                                                              \
                        const long long si = ((long long)n * Lq + q) * M + m;                                                  \
                        const T* wp = weight + si * L * P;                                                                     \
                        const T* lp = loc + si * L * P * 2;                                                                    \
                        T col = 0;                                                                                             \
                        for (int l = 0; l < L; ++l) {                                                                          \
                            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                      \
                            const T* v = value + ((long long)n * S + level_start[l]) * M * D;                                  \
                            for (int p = 0; p < P; ++p) {                                                                      \
                                const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];                          \
                                const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;                                  \
                                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                                            \
                                    col += NAME##_bilinear(v, H, W, M, D, h_im, w_im, m, c) * wp[l * P + p];                   \
                            }                                                                                                  \
                        }                                                                                                      \
                        out[si * D + c] = col;                                                                                 \
                    }                                                                                                          \
    }

MSDA_IMPL(msda_forward_f32, float, floorf)
MSDA_IMPL(msda_forward_f64, double, floor)
