/* A plain-C host that DRIVES the hot path on the GPU without Python or torch: it owns the device memory (HIP runtime
 * C API), dlopen()s libfo1hip.so and launches
 *   1. fo1_hfre_region_pool_ex   — three boxes on a two-level aux pyramid of constant maps: the 7x7 ROI mean of a constant map is that
 *      constant (whatever the box, incl. one hanging over the border inside the map's [-1, H] sampling band), the bf16 second destination is its
 *      RNE cast; replaces HFREModule.__call__ (hybrid_finegrained_region_encoder.py:275-469);
 *   2. fo1_llm_prefill            — a one-layer Qwen2.5 decoder (hidden 256, 2 q heads / 1 kv head, head_dim 128) over one 8-row prompt
 *      with zero projection weights: attention and MLP then add exactly 0 to the residual stream, so the last hidden row is
 *      RMSNorm(embeds[7]) and, with lm_head row j = unit vector e_(j mod 256), the greedy id is the column of the row's maximum;
 *      replaces Qwen2_5_VLModel.forward + lm_head + argmax (modeling_qwen2_5_vl.py:1126-1242, omchat_qwen2_5_vl.py:38).
 * Built (gcc + libamdhip64) and run by tests/test_gpu_c_host.py on the GPU box.  usage: gpu_host /path/to/libfo1hip.so */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "fo1.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 10; } } while (0)

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

typedef const char* (*last_error_fn)(void);
typedef size_t (*hfre_ws_fn)(const fo1_hfre_source_t*, int, int);
typedef int (*hfre_fn)(const fo1_hfre_source_t*, int, const float*, int, const float*, float, float, int, int, float, float, float*, int, int,
                       const fo1_hfre_opts_t*, void*, size_t, void*);
typedef size_t (*llm_ws_fn)(const fo1_llm_weights_t*, int, int);
typedef int (*llm_fn)(const fo1_llm_weights_t*, const fo1_kv_cache_t*, const void*, int, const void*, const void*, int, int, const int32_t*, int, int,
                      double, const int32_t*, int, void*, void*, void*, int32_t*, void*, size_t, void*);

static void* dev_upload(const void* src, size_t n) {
    void* d = NULL;
    if (hipMalloc(&d, n ? n : 16) != hipSuccess) return NULL;
    if (src) { if (hipMemcpy(d, src, n, hipMemcpyHostToDevice) != hipSuccess) return NULL; }
    else if (hipMemset(d, 0, n ? n : 16) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: gpu_host libfo1hip.so\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    last_error_fn last_error = (last_error_fn)dlsym(h, "fo1_last_error");
    hfre_ws_fn hfre_ws = (hfre_ws_fn)dlsym(h, "fo1_hfre_ex_workspace_bytes");
    hfre_fn hfre = (hfre_fn)dlsym(h, "fo1_hfre_region_pool_ex");
    llm_ws_fn llm_ws = (llm_ws_fn)dlsym(h, "fo1_llm_prefill_workspace_bytes");
    llm_fn llm = (llm_fn)dlsym(h, "fo1_llm_prefill");
    if (!last_error || !hfre_ws || !hfre || !llm_ws || !llm) { fprintf(stderr, "missing symbol\n"); return 4; }
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    /* ---------------- 1. HFRE region pool ---------------- */
    {
        enum { H0 = 24, W0 = 32, C0 = 64, H1 = 12, W1 = 16, C1 = 128, NB = 3, RD = C0 + C1 };
        const float v0 = 0.75f, v1 = -1.5f;      /* exactly representable in bf16 */
        uint16_t* m0 = (uint16_t*)malloc(sizeof(uint16_t) * H0 * W0 * C0);
        uint16_t* m1 = (uint16_t*)malloc(sizeof(uint16_t) * H1 * W1 * C1);
        for (int i = 0; i < H0 * W0 * C0; ++i) m0[i] = bf16(v0);
        for (int i = 0; i < H1 * W1 * C1; ++i) m1[i] = bf16(v1);
        const float boxes[NB * 4] = {8.f, 8.f, 60.f, 40.f, 0.f, 0.f, 127.f, 95.f, 100.f, 70.f, 128.f, 96.f};   /* aux px, scale 0.25 */
        void *d0 = dev_upload(m0, sizeof(uint16_t) * H0 * W0 * C0), *d1 = dev_upload(m1, sizeof(uint16_t) * H1 * W1 * C1);
        float* dbox = (float*)dev_upload(boxes, sizeof boxes);
        float* dout = (float*)dev_upload(NULL, sizeof(float) * NB * RD);
        uint16_t* dout16 = (uint16_t*)dev_upload(NULL, sizeof(uint16_t) * NB * RD);
        if (!d0 || !d1 || !dbox || !dout || !dout16) return 11;
        fo1_hfre_source_t src[2];
        memset(src, 0, sizeof src);
        src[0].data = d0; src[0].H = H0; src[0].W = W0; src[0].C = C0; src[0].ld = C0; src[0].roi_H = H0; src[0].roi_W = W0;
        src[0].spatial_scale = 0.25f; src[0].box_space = 0; src[0].out_offset = 0;
        src[1].data = d1; src[1].H = H1; src[1].W = W1; src[1].C = C1; src[1].ld = C1; src[1].roi_H = H0; src[1].roi_W = W0;   /* upsampled to level 0 */
        src[1].spatial_scale = 0.25f; src[1].box_space = 0; src[1].out_offset = C0;
        fo1_hfre_opts_t opts;
        memset(&opts, 0, sizeof opts);
        opts.batch = 1; opts.ln_eps = 1e-5f; opts.out_bf16 = dout16; opts.out_bf16_ld = RD;
        const size_t need = hfre_ws(src, 2, NB);
        void* ws = dev_upload(NULL, need);
        if (!ws) return 12;
        int rc = hfre(src, 2, dbox, NB, NULL, 1.f, 1.f, 7, 0 /* no position embedding */, 1.f, 1.f, dout, RD, RD, &opts, ws, need, (void*)st);
        if (rc != 0) { fprintf(stderr, "fo1_hfre_region_pool_ex rc=%d: %s\n", rc, last_error()); return 13; }
        CK(hipStreamSynchronize(st));
        float out[NB * RD];
        uint16_t out16[NB * RD];
        CK(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
        CK(hipMemcpy(out16, dout16, sizeof out16, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int n = 0; n < NB; ++n)
            for (int c = 0; c < RD; ++c) {
                const double want = c < C0 ? v0 : v1, d = fabs(out[n * RD + c] - want);
                if (d > worst) worst = d;
                if (out16[n * RD + c] != bf16(out[n * RD + c])) { fprintf(stderr, "hfre: bf16 destination is not the RNE cast at [%d,%d]\n", n, c); return 14; }
            }
        if (worst > 2e-5) { fprintf(stderr, "hfre: constant maps pooled to %g off\n", worst); return 15; }
        printf("hfre ok: %d boxes x %d channels, max |err| %.3g, bf16 destination = RNE cast\n", NB, RD, worst);
        /* argument error: never throws across the ABI, message retrievable */
        rc = hfre(src, 2, dbox, NB, NULL, 1.f, 1.f, 7, 0, 1.f, 1.f, dout, RD, RD - 8, &opts, ws, need, (void*)st);
        if (rc >= 0 || strlen(last_error()) == 0) { fprintf(stderr, "hfre: bad region_dim was not refused\n"); return 16; }
    }

    /* ---------------- 2. one-layer LLM prefill ---------------- */
    {
        enum { HID = 256, NH = 2, NKV = 1, HD = 128, FF = 512, V = 1024, R = 8, CAP = 64 };
        const size_t qkv_rows = (NH + 2 * NKV) * HD;
        uint16_t* ones = (uint16_t*)malloc(sizeof(uint16_t) * HID);
        for (int i = 0; i < HID; ++i) ones[i] = bf16(1.0f);
        void* d_ones = dev_upload(ones, sizeof(uint16_t) * HID);
        void* d_wqkv = dev_upload(NULL, sizeof(uint16_t) * qkv_rows * HID);
        void* d_bqkv = dev_upload(NULL, sizeof(uint16_t) * qkv_rows);
        void* d_wo = dev_upload(NULL, sizeof(uint16_t) * HID * NH * HD);
        void* d_wgu = dev_upload(NULL, sizeof(uint16_t) * 2 * FF * HID);
        void* d_wdown = dev_upload(NULL, sizeof(uint16_t) * HID * FF);
        uint16_t* head = (uint16_t*)calloc((size_t)V * HID, sizeof(uint16_t));
        for (int j = 0; j < V; ++j) head[(size_t)j * HID + (j % HID)] = bf16(1.0f);
        void* d_head = dev_upload(head, sizeof(uint16_t) * V * HID);
        uint16_t* emb = (uint16_t*)malloc(sizeof(uint16_t) * R * HID);
        int want_col = 0;
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < HID; ++c) {
                float v = 0.01f * (float)(((r * 131 + c * 17) % 97) - 48);
                if (r == R - 1 && c == 77) v = 3.0f;          /* the last row's maximum */
                emb[r * HID + c] = bf16(v);
            }
        want_col = 77;
        void* d_emb = dev_upload(emb, sizeof(uint16_t) * R * HID);
        uint16_t* cs = (uint16_t*)malloc(sizeof(uint16_t) * R * HD);
        for (int i = 0; i < R * HD; ++i) cs[i] = bf16(1.0f);
        void* d_cos = dev_upload(cs, sizeof(uint16_t) * R * HD);
        void* d_sin = dev_upload(NULL, sizeof(uint16_t) * R * HD);
        const int32_t items[4] = {0, R, 0, R}, last_plan[2] = {0, R - 1};
        void* d_items = dev_upload(items, sizeof items);
        void* d_last = dev_upload(last_plan, sizeof last_plan);
        void* d_k = dev_upload(NULL, sizeof(uint16_t) * NKV * CAP * HD);
        void* d_vt = dev_upload(NULL, sizeof(uint16_t) * NKV * HD * CAP);
        void* d_lasth = dev_upload(NULL, sizeof(uint16_t) * HID);
        void* d_logits = dev_upload(NULL, sizeof(uint16_t) * V);
        int32_t* d_next = (int32_t*)dev_upload(NULL, sizeof(int32_t));
        if (!d_ones || !d_wqkv || !d_bqkv || !d_wo || !d_wgu || !d_wdown || !d_head || !d_emb || !d_cos || !d_sin || !d_items || !d_last || !d_k || !d_vt ||
            !d_lasth || !d_logits || !d_next) return 21;
        fo1_llm_layer_t layer;
        layer.ln1 = d_ones; layer.ln2 = d_ones; layer.wqkv = d_wqkv; layer.bqkv = d_bqkv; layer.wo = d_wo; layer.wgu = d_wgu; layer.wdown = d_wdown;
        fo1_llm_weights_t w;
        memset(&w, 0, sizeof w);
        w.n_layers = 1; w.hidden = HID; w.n_heads = NH; w.n_kv_heads = NKV; w.head_dim = HD; w.intermediate = FF; w.vocab = V; w.rms_eps = 1e-6f;
        w.layers = &layer; w.embed = d_head; w.final_norm = d_ones; w.lm_head = d_head;
        fo1_kv_cache_t kv;
        kv.k = d_k; kv.k_layer_stride = (long long)NKV * CAP * HD; kv.k_head_stride = (long long)CAP * HD;
        kv.vt = d_vt; kv.vt_layer_stride = (long long)NKV * HD * CAP; kv.vt_row_stride = CAP; kv.capacity = CAP;
        const size_t need = llm_ws(&w, R, 1);
        void* ws = dev_upload(NULL, need);
        if (!ws) return 22;
        int rc = llm(&w, &kv, d_emb, HID, d_cos, d_sin, R, 0, (const int32_t*)d_items, 1, 64, 4.0 * NH * HD * R * (R + 1) / 2.0, (const int32_t*)d_last, 1,
                     NULL, d_lasth, d_logits, d_next, ws, need, (void*)st);
        if (rc != 0) { fprintf(stderr, "fo1_llm_prefill rc=%d: %s\n", rc, last_error()); return 23; }
        CK(hipStreamSynchronize(st));
        int32_t next = -1;
        uint16_t lasth[HID];
        CK(hipMemcpy(&next, d_next, sizeof next, hipMemcpyDeviceToHost));
        CK(hipMemcpy(lasth, d_lasth, sizeof lasth, hipMemcpyDeviceToHost));
        /* RMSNorm(embeds[R-1]) with unit weight, the reference's rounding points: fp32 variance, bf16(x * rstd) (modeling_qwen2_5_vl.py:126-140) */
        double ss = 0;
        for (int c = 0; c < HID; ++c) { const double x = f32(emb[(R - 1) * HID + c]); ss += x * x; }
        const float rstd = 1.0f / sqrtf((float)(ss / HID) + 1e-6f);
        double worst = 0;
        for (int c = 0; c < HID; ++c) {
            const float ref = f32(bf16(f32(emb[(R - 1) * HID + c]) * rstd));
            const double d = fabs(f32(lasth[c]) - ref);
            if (d > worst) worst = d;
        }
        if (worst > 1.0 / 64) { fprintf(stderr, "llm: last hidden row off by %g\n", worst); return 24; }
        if (next != want_col) { fprintf(stderr, "llm: greedy id %d, expected %d\n", next, want_col); return 25; }
        printf("llm ok: one-layer prefill of %d rows, last hidden max |err| %.3g, greedy id %d\n", R, worst, next);
    }
    printf("gpu_host ok\n");
    return 0;
}
