/*
 * ORACLE — test infrastructure only.  Never imported by the product path
 * (vlm_fo1_amd/, vlm_fo1/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * CPU restatement of torchvision.ops.roi_align (torchvision==0.21.0, pinned
 * at /root/reference/requirements.txt:2; the source is NOT vendored in the
 * reference tree).  The reference calls it at
 *   vlm_fo1/model/multimodal_visual_prompt_encoder/hybrid_finegrained_region_encoder.py:248,263,353
 * with output_size=7, sampling_ratio=-1 (adaptive), aligned=False, fp32 NCHW
 * input and a single-image box list.  This file restates the published CPU
 * algorithm (ROIAlign forward with pre-computed bilinear taps) in float
 * arithmetic, then folds in the spatial mean over the pooled bins that HFRE
 * applies right after (`.mean(dim=(2, 3))`, same file :255,270,361).
 *
 * Parity status: "parity unpinned" for roi_align itself — the reference tree
 * holds no golden vector for it and torchvision is not installed here; the
 * restatement is cross-checked against an independent torch restatement
 * (oracle/hfre_oracle.py) and, through it, against the reference's own
 * HFREModule run in this container with this function injected.
 *
 * Layout: input is read through explicit strides so both true NCHW tensors
 * and the reference's channels-last views (DaViT / ViT maps) work.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int pos1, pos2, pos3, pos4;
    float w1, w2, w3, w4;
} tap_t;

/* Full roi_align: out[n, c, ph, pw] (contiguous). */
void oracle_roi_align_f32(const float* in, int64_t C, int64_t H, int64_t W,
                          int64_t sc, int64_t sh, int64_t sw, /* element strides */
                          const float* rois, int64_t N,      /* [N,4] xyxy */
                          float spatial_scale, int pooled, int sampling_ratio, int aligned,
                          float* out)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t n = 0; n < N; ++n) {
        const float* r = rois + 4 * n;
        const float offset = aligned ? 0.5f : 0.0f;
        float roi_start_w = r[0] * spatial_scale - offset;
        float roi_start_h = r[1] * spatial_scale - offset;
        float roi_end_w = r[2] * spatial_scale - offset;
        float roi_end_h = r[3] * spatial_scale - offset;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        if (!aligned) {
            roi_width = fmaxf(roi_width, 1.0f);
            roi_height = fmaxf(roi_height, 1.0f);
        }
        float bin_h = roi_height / (float)pooled;
        float bin_w = roi_width / (float)pooled;
        int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled);
        int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled);
        float count = (float)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);

        size_t ntaps = (size_t)pooled * pooled * grid_h * grid_w;
        tap_t* taps = (tap_t*)malloc(sizeof(tap_t) * (ntaps ? ntaps : 1));
        size_t t = 0;
        for (int ph = 0; ph < pooled; ++ph)
            for (int pw = 0; pw < pooled; ++pw)
                for (int iy = 0; iy < grid_h; ++iy) {
                    const float yy = roi_start_h + ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
                    for (int ix = 0; ix < grid_w; ++ix) {
                        const float xx = roi_start_w + pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
                        float x = xx, y = yy;
                        tap_t tp;
                        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
                            memset(&tp, 0, sizeof tp);
                            taps[t++] = tp;
                            continue;
                        }
                        if (y <= 0) y = 0;
                        if (x <= 0) x = 0;
                        int y_low = (int)y, x_low = (int)x, y_high, x_high;
                        if (y_low >= H - 1) { y_high = y_low = (int)H - 1; y = (float)y_low; } else y_high = y_low + 1;
                        if (x_low >= W - 1) { x_high = x_low = (int)W - 1; x = (float)x_low; } else x_high = x_low + 1;
                        float ly = y - y_low, lx = x - x_low, hy = 1.0f - ly, hx = 1.0f - lx;
                        tp.pos1 = (int)(y_low * sh + x_low * sw);
                        tp.pos2 = (int)(y_low * sh + x_high * sw);
                        tp.pos3 = (int)(y_high * sh + x_low * sw);
                        tp.pos4 = (int)(y_high * sh + x_high * sw);
                        tp.w1 = hy * hx; tp.w2 = hy * lx; tp.w3 = ly * hx; tp.w4 = ly * lx;
                        taps[t++] = tp;
                    }
                }
        for (int64_t c = 0; c < C; ++c) {
            const float* base = in + c * sc;
            size_t k = 0;
            for (int ph = 0; ph < pooled; ++ph)
                for (int pw = 0; pw < pooled; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < grid_h; ++iy)
                        for (int ix = 0; ix < grid_w; ++ix) {
                            const tap_t tp = taps[k++];
                            acc += tp.w1 * base[tp.pos1] + tp.w2 * base[tp.pos2] +
                                   tp.w3 * base[tp.pos3] + tp.w4 * base[tp.pos4];
                        }
                    out[((n * C + c) * pooled + ph) * pooled + pw] = acc / count;
                }
        }
        free(taps);
    }
}

/* roi_align followed by the spatial mean HFRE applies: out[n, c]. */
void oracle_roi_align_mean_f32(const float* in, int64_t C, int64_t H, int64_t W,
                               int64_t sc, int64_t sh, int64_t sw,
                               const float* rois, int64_t N,
                               float spatial_scale, int pooled, int sampling_ratio, int aligned,
                               float* out)
{
    float* full = (float*)malloc(sizeof(float) * (size_t)N * C * pooled * pooled);
    oracle_roi_align_f32(in, C, H, W, sc, sh, sw, rois, N, spatial_scale, pooled, sampling_ratio, aligned, full);
    const int pp = pooled * pooled;
    for (int64_t i = 0; i < N * C; ++i) {
        /* torch .mean over 49 fp32 values: pairwise/vectorised in torch; plain
         * sequential sum here (difference is O(1e-7) relative). */
        float s = 0.0f;
        for (int k = 0; k < pp; ++k) s += full[i * pp + k];
        out[i] = s / (float)pp;
    }
    free(full);
}
