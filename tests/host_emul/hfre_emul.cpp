// Host-side unit harness for vlm_fo1_amd/csrc/hfre_math.h (TEST CODE ONLY — never
// part of the product path).  Executes the exact per-axis weight / footprint / slice
// functions the HIP kernel uses, with the kernel's loop structure flattened to plain
// loops, so the index math can be checked against the oracle on a machine with no GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../vlm_fo1_amd/csrc/hfre_math.h"

using namespace fo1;

extern "C" {

struct emul_source {
    const float* data;  // channels-last [H*W, ld] fp32 (bf16-valued in the tests)
    int32_t H, W, C, ld, roi_H, roi_W;
    float spatial_scale;
    int32_t box_space, out_offset;
};

// out[n, out_offset + c] (no position embedding); returns number of slices processed.
long hfre_emul_pool(const emul_source* srcs, int n_sources, const float* boxes, int n_boxes, float vsx, float vsy,
                    int P, int pixel_budget, float* out, int out_ld) {
    long total_slices = 0;
    for (int si = 0; si < n_sources; ++si) {
        const emul_source& s = srcs[si];
        for (int n = 0; n < n_boxes; ++n) {
            float x1 = boxes[4 * n], y1 = boxes[4 * n + 1], x2 = boxes[4 * n + 2], y2 = boxes[4 * n + 3];
            if (s.box_space == 1) { x1 *= vsx; x2 *= vsx; y1 *= vsy; y2 *= vsy; }
            RoiAxis ay = make_roi_axis(y1, y2, s.spatial_scale, P, s.roi_H);
            RoiAxis ax = make_roi_axis(x1, x2, s.spatial_scale, P, s.roi_W);
            int r_lo, r_hi, c_lo, c_hi;
            upsample_range(ay.lo, ay.hi, s.H, s.roi_H, r_lo, r_hi);
            upsample_range(ax.lo, ax.hi, s.W, s.roi_W, c_lo, c_hi);
            std::vector<double> acc(s.C, 0.0);
            const int fh = r_hi - r_lo + 1, fw = c_hi - c_lo + 1;
            if (fh > 0 && fw > 0) {
                std::vector<float> wAy(ay.hi >= ay.lo ? ay.hi - ay.lo + 1 : 0), wAx(ax.hi >= ax.lo ? ax.hi - ax.lo + 1 : 0);
                for (int a = ay.lo; a <= ay.hi; ++a) wAy[a - ay.lo] = roi_axis_weight(ay, a);
                for (int a = ax.lo; a <= ax.hi; ++a) wAx[a - ax.lo] = roi_axis_weight(ax, a);
                const int R = slice_rows(fw, pixel_budget);
                const int nsl = (fh + R - 1) / R;
                const int max_slices = (s.H + slice_rows(s.W, pixel_budget) - 1) / slice_rows(s.W, pixel_budget);
                if (nsl > max_slices) return -1;  // the grid bound the kernel relies on
                std::vector<float> wx(fw);
                for (int c = 0; c < fw; ++c)
                    wx[c] = (s.W != s.roi_W) ? upsample_axis_weight(c_lo + c, wAx.data(), ax.lo, ax.hi, s.W, s.roi_W)
                                             : roi_axis_weight(ax, c_lo + c);
                for (int k = 0; k < nsl; ++k) {
                    ++total_slices;
                    const int row0 = r_lo + k * R;
                    int row1 = row0 + R - 1;
                    if (row1 > r_hi) row1 = r_hi;
                    std::vector<float> part(s.C, 0.0f);
                    for (int r = row0; r <= row1; ++r) {
                        const float wy = (s.H != s.roi_H) ? upsample_axis_weight(r, wAy.data(), ay.lo, ay.hi, s.H, s.roi_H)
                                                          : roi_axis_weight(ay, r);
                        for (int c = 0; c < fw; ++c) {
                            const float w = wy * wx[c];
                            const float* px = s.data + ((size_t)r * s.W + (size_t)(c_lo + c)) * s.ld;
                            for (int ch = 0; ch < s.C; ++ch) part[ch] += w * px[ch];
                        }
                    }
                    for (int ch = 0; ch < s.C; ++ch) acc[ch] += part[ch];
                }
            }
            for (int ch = 0; ch < s.C; ++ch) out[(size_t)n * out_ld + s.out_offset + ch] = (float)acc[ch];
        }
    }
    return total_slices;
}

// Sum of all per-axis weights of one ROI axis (must be 1 when every sample is valid)
// and the weights themselves for inspection.
int hfre_emul_axis(float lo, float hi, float scale, int P, int L, float* w_out /*[L]*/) {
    RoiAxis a = make_roi_axis(lo, hi, scale, P, L);
    for (int i = 0; i < L; ++i) w_out[i] = roi_axis_weight(a, i);
    return a.hi - a.lo + 1;
}

}  // extern "C"
