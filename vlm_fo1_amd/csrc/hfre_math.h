// hfre_math.h — per-axis ROI-align / upsample tap-weight math shared by the HIP
// kernels (hfre.hip) and the host-side unit harness (tests/host_emul/).
//
// roi_align semantics follow torchvision 0.21.0 (aligned=False, adaptive
// sampling_ratio) as called from the reference's
//   hybrid_finegrained_region_encoder.py:248,263,353
// and the bilinear upsample follows torch F.interpolate(align_corners=False)
// called at :341-346.  mean_{PxP}(roi_align(...)) factorises exactly into
//   out[c] = sum_h sum_w wy[h] * wx[w] * F[h, w, c]
// because every rule (validity window, clamping, top-edge collapse, 1/grid
// normalisation) is per-axis.  All coordinate arithmetic is fp32, evaluated with
// the same expression shapes as the reference so floor() decisions agree.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FO1_HD __host__ __device__ __forceinline__
#else
#define FO1_HD inline
#endif

namespace fo1 {

// One axis of one ROI on the roi map (length L).
struct RoiAxis {
    float start;  // roi_start = box_lo * spatial_scale
    float bin;    // roi_len / P, roi_len = max(box_hi*s - box_lo*s, 1)
    int grid;     // ceil(roi_len / P)
    int P;        // pooled bins
    int L;        // roi-map extent along this axis
    int lo, hi;   // inclusive tap range on the roi map (hi < lo  => empty)
};

FO1_HD float roi_sample_coord(const RoiAxis& a, int s) {
    const int ph = s / a.grid;
    const int i = s - ph * a.grid;
    return a.start + (float)ph * a.bin + ((float)i + 0.5f) * a.bin / (float)a.grid;
}

FO1_HD RoiAxis make_roi_axis(float box_lo, float box_hi, float spatial_scale, int P, int L) {
    RoiAxis a;
    const float s0 = box_lo * spatial_scale;
    const float s1 = box_hi * spatial_scale;
    float len = s1 - s0;
    len = fmaxf(len, 1.0f);
    a.start = s0;
    a.bin = len / (float)P;
    a.grid = (int)ceilf(len / (float)P);
    if (a.grid < 1) a.grid = 1;
    a.P = P;
    a.L = L;
    // tap range from the first / last sample (samples are monotone in s)
    const float vf = roi_sample_coord(a, 0);
    const float vl = roi_sample_coord(a, P * a.grid - 1);
    if (!(vl >= -1.0f) || !(vf <= (float)L) || !(len < 1e8f)) {  // nothing valid (or NaN / absurd box)
        a.lo = 0;
        a.hi = -1;
        return a;
    }
    int lo = vf <= 0.0f ? 0 : (int)fminf(vf, (float)(L - 1));
    int hi = vl >= (float)(L - 1) ? L - 1 : (vl <= 0.0f ? 0 : (int)vl) + 1;
    if (hi > L - 1) hi = L - 1;
    if (lo > L - 1) lo = L - 1;
    if (hi < lo) hi = lo;
    a.lo = lo;
    a.hi = hi;
    return a;
}

// Weight of roi-map index `idx` for this axis: sum over the axis' P*grid samples of
// the bilinear tap weight landing on idx, times 1/(grid*P) (bin average and the
// spatial mean over the P bins).
FO1_HD float roi_axis_weight(const RoiAxis& a, int idx) {
    if (idx < a.lo || idx > a.hi) return 0.0f;
    const int S = a.P * a.grid;
    const float d = a.bin / (float)a.grid;  // sample spacing (<= 1)
    const float vmin = (idx == 0) ? -1.0f : (float)(idx - 1);
    const float vmax = (idx == a.L - 1) ? (float)a.L : (float)(idx + 1);
    float flo = floorf((vmin - a.start) / d - 0.5f) - 1.0f;
    float fhi = ceilf((vmax - a.start) / d - 0.5f) + 1.0f;
    flo = fminf(fmaxf(flo, 0.0f), (float)(S - 1));
    fhi = fminf(fmaxf(fhi, 0.0f), (float)(S - 1));
    const int s_lo = (int)flo, s_hi = (int)fhi;
    float acc = 0.0f;
    for (int s = s_lo; s <= s_hi; ++s) {
        float v = roi_sample_coord(a, s);
        if (v < -1.0f || v > (float)a.L) continue;
        if (v <= 0.0f) v = 0.0f;
        int l = (int)v, h;
        if (l >= a.L - 1) {
            h = l = a.L - 1;
            v = (float)l;
        } else {
            h = l + 1;
        }
        const float wl = v - (float)l;
        const float wh = 1.0f - wl;
        if (l == idx) acc += wh;
        if (h == idx) acc += wl;
    }
    return acc / (float)S;
}

// Bilinear upsample (align_corners=False) source index/lambda of output index o.
FO1_HD void upsample_src(int o, int Lin, int Lout, int& i0, int& i1, float& l0, float& l1) {
    if (Lin == Lout) {
        i0 = i1 = o;
        l0 = 1.0f;
        l1 = 0.0f;
        return;
    }
    const float scale = (float)Lin / (float)Lout;
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int f = (int)floorf(src);
    if (f > Lin - 1) f = Lin - 1;
    float lam = src - (float)f;
    lam = fminf(fmaxf(lam, 0.0f), 1.0f);
    i0 = f;
    i1 = f + ((f < Lin - 1) ? 1 : 0);
    l1 = lam;
    l0 = 1.0f - lam;
}

// Inclusive range of source indices touched by roi-map indices [lo, hi].
FO1_HD void upsample_range(int lo, int hi, int Lin, int Lout, int& r_lo, int& r_hi) {
    if (hi < lo) {
        r_lo = 0;
        r_hi = -1;
        return;
    }
    int i0, i1;
    float l0, l1;
    upsample_src(lo, Lin, Lout, i0, i1, l0, l1);
    r_lo = i0;
    upsample_src(hi, Lin, Lout, i0, i1, l0, l1);
    r_hi = i1;
}

// Weight of source index r: sum_a wA[a] * U[a, r] with wA given on [a_lo, a_hi]
// (wA[0] is the weight of roi-map index a_lo).
FO1_HD float upsample_axis_weight(int r, const float* wA, int a_lo, int a_hi, int Lin, int Lout) {
    if (Lin == Lout) return (r >= a_lo && r <= a_hi) ? wA[r - a_lo] : 0.0f;
    const float inv = (float)Lout / (float)Lin;
    // src(a) in (r-1, r+1)  =>  a in ((r-0.5)*inv-0.5, (r+1.5)*inv-0.5); widen by 1 each side
    float flo = floorf(((float)r - 0.5f) * inv - 0.5f) - 1.0f;
    float fhi = ceilf(((float)r + 1.5f) * inv - 0.5f) + 1.0f;
    int c_lo = (r == 0) ? a_lo : (int)fmaxf(flo, (float)a_lo);
    int c_hi = (int)fminf(fhi, (float)a_hi);
    if (c_lo < a_lo) c_lo = a_lo;
    float acc = 0.0f;
    for (int a = c_lo; a <= c_hi; ++a) {
        int i0, i1;
        float l0, l1;
        upsample_src(a, Lin, Lout, i0, i1, l0, l1);
        float u = 0.0f;
        if (i0 == r) u += l0;
        if (i1 == r) u += l1;
        acc += wA[a - a_lo] * u;
    }
    return acc;
}

// Rows per slice for a footprint `fw` pixels wide under a per-workgroup pixel budget.
FO1_HD int slice_rows(int fw, int pixel_budget) {
    int r = pixel_budget / (fw > 0 ? fw : 1);
    return r < 1 ? 1 : r;
}

}  // namespace fo1
