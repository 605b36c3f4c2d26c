/*
 * fo1.h — C-ABI of libfo1hip.so, the MI355X (gfx950) engine for the VLM-FO1 hot path.
 *
 * Plain C: raw device pointers, explicit shapes/strides, an opaque stream handle.
 * No torch / HIP types in any signature.  Every entry point returns int:
 *   0 = ok, <0 = argument/shape error (text via fo1_last_error()), >0 = hipError_t.
 * The library never allocates or frees device memory: outputs and workspaces are
 * caller-owned (PyTorch's caching allocator in the Python host).  Inputs are const.
 * All kernels are enqueued asynchronously on `stream` (a hipStream_t cast to void*;
 * NULL = the legacy default stream).
 *
 * Each function cites the reference interface it replaces (paths relative to
 * om-ai-lab/VLM-FO1 @ 2025-10-31).  INTEGRATION.md shows the ctypes binding the
 * reference-side maintainer would add.
 */
#ifndef FO1_H
#define FO1_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO1_ABI_VERSION 1
#define FO1_OK 0
#define FO1_ERR_ARG (-1)       /* bad argument / unsupported shape */
#define FO1_ERR_WORKSPACE (-2) /* workspace too small */

int fo1_abi_version(void);
/* Thread-local text of the last error returned on this thread ("" if none). */
const char* fo1_last_error(void);

/* ------------------------------------------------------------------------
 * Per-kernel timing for bench.py's roofline line.  While enabled, every kernel the
 * library launches is bracketed by hipEvents on its launch stream.  Off by default and
 * during throughput timing (the event records perturb back-to-back launches).
 * `total_work` accumulates the ALGORITHMIC bytes (HBM-bound kernels) or flops
 * (MFMA-bound kernels) of each launch, as documented per kernel in DESIGN.md.
 * ---------------------------------------------------------------------- */
typedef struct fo1_profile_row {
    char name[48];
    int64_t calls;
    double total_ms;
    double total_work;
} fo1_profile_row_t;
int fo1_profile_enable(int on);
int fo1_profile_read(fo1_profile_row_t* rows, int cap, int reset);

/* ------------------------------------------------------------------------
 * HFRE region pooling  
 *
 * Replaces the arithmetic of
 *   HFREModule.__call__                 hybrid_finegrained_region_encoder.py:275-469
 *   HFREModule.extract_vt_region_feature                                   :230-273
 *   gen_sineembed_for_position                                             :55-103
 *   the three torchvision.ops.roi_align call sites                         :248,263,353
 *   F.interpolate(bilinear) + torch.cat of the aux pyramid                 :338-350
 *   the aux→vt box scaling in encode_regions      omchat_qwen2_5_vl.py:94-99
 * for the product configuration (use_vision_tower_region_feature, 'concat',
 * 'bbox_based').  One call = all boxes of one image, all feature sources.
 *
 * out[n, src.out_offset + c] =
 *     mean_{7x7 bins}( roi_align(src map (bilinearly upsampled to roi_H x roi_W when
 *                       they differ), box_n * src.spatial_scale, aligned=False,
 *                       sampling_ratio=-1) )[c]
 *   + sine box embedding (pos_mode != 0)
 * computed with the exact separable form (per-axis tap weights composed with the
 * per-axis upsample matrix), so each source map is read once at native
 * resolution in bf16 and no [C,H0,W0] fp32 intermediate exists.
 * ---------------------------------------------------------------------- */
#define FO1_HFRE_MAX_SOURCES 8
#define FO1_HFRE_MAX_EXTENT 1024 /* max H or W of any source / roi map */

typedef struct fo1_hfre_source {
    const void* data;    /* device, bf16, token-major (channels-last): element (h,w,c) at
                            data[(h*W + w)*ld + c]                                        */
    int32_t H, W;        /* map size                                                       */
    int32_t C;           /* channels pooled from this source (multiple of 64)              */
    int32_t ld;          /* row stride in elements (>= C, multiple of 8; data 16-B aligned)*/
    int32_t roi_H, roi_W;/* size of the map roi_align runs on; != (H,W) means the source is
                            bilinearly upsampled (align_corners=False) to it first         */
    float spatial_scale; /* roi_align spatial_scale on the roi_H x roi_W map               */
    int32_t box_space;   /* 0: aux boxes as given; 1: vt boxes = aux * (vt_scale_x, _y)    */
    int32_t out_offset;  /* first output channel written by this source                    */
} fo1_hfre_source_t;

/* Tuning hook: footprint pixels one workgroup streams per row-slice (default 1024). */
int fo1_hfre_set_pixel_budget(int pixels);

/* Bytes of scratch fo1_hfre_region_pool needs for these sources / n_boxes. */
size_t fo1_hfre_workspace_bytes(const fo1_hfre_source_t* sources, int n_sources, int n_boxes);

int fo1_hfre_region_pool(
    const fo1_hfre_source_t* sources, int n_sources, /* host array, copied at launch */
    const float* boxes_aux, int n_boxes,             /* device fp32 [n_boxes,4] xyxy, aux px */
    const float* boxes_vt,                           /* device fp32 [n_boxes,4] in vt px, or NULL:
                                                        then vt box = aux box * (vt_scale_x,_y),
                                                        one fp32 multiply as omchat_qwen2_5_vl.py:99 */
    float vt_scale_x, float vt_scale_y,
    int roi_size,                                    /* pooled bins per axis (7)             */
    int pos_mode,                                    /* 0 none; 1 embed vt boxes; 2 aux boxes*/
    float pos_img_w, float pos_img_h,                /* box normalisers (grid*14 / aux size) */
    float* out, int out_ld, int region_dim,          /* device fp32 [n_boxes, out_ld]        */
    void* workspace, size_t workspace_bytes,         /* device scratch                       */
    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FO1_H */
