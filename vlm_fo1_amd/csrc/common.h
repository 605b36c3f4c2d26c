// common.h — shared helpers for the libfo1hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fo1.h"
#include "ab.h"   // test / bench build: include/fo1_ab.h (its declarations carry the export visibility)

namespace fo1 {

// thread-local error text (fo1_last_error)
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define FO1_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return fo1::set_err(FO1_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define FO1_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fo1::set_err((int)_e, "%s failed: %s", #expr, hipGetErrorString(_e));     \
    } while (0)

#define FO1_CHECK_LAUNCH()                                                                        \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess)                                                                     \
            return fo1::set_err((int)_e, "kernel launch failed at %s:%d: %s", __FILE__, __LINE__, \
                                hipGetErrorString(_e));                                           \
    } while (0)

// bf16 <-> f32 bit helpers (round-to-nearest-even on the way down, like torch)
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction where the bit-twiddled form costs six
// and a branch; every epilogue in the library goes through these two.
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float fo1_f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 fo1_bf16x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(fo1_f32x2{lo, hi}, fo1_bf16x2));
}

// Activations of the fused epilogues.  Their result is rounded to bf16 (8 significant bits) right away, so the hardware
// approximations (v_exp_f32, v_rcp_f32: ~1 ulp of fp32) replace the library's expf / erff / IEEE division — ~6 instructions per
// element instead of ~25-40, in epilogues that run 128 elements per lane (DaViT fc1 + GELU ran at 564 TFLOP/s for that reason).
// SiLU (Qwen2MLP act_fn, modeling_qwen2_5_vl.py:636):  v * sigmoid(v).
__device__ __forceinline__ float fo1_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// erf-GELU (nn.GELU(): modeling_davit.py:57, simple_fpn.py:145, multimodal_projector/builder.py:69,108, merger): 0.5 v (1 + erf(v / sqrt 2))
// with erfc(|x|) from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute); 1 + erf(x) is formed as erfc(|x|) on the negative
// side (no cancellation) and 2 - erfc(x) on the positive side.
__device__ __forceinline__ float fo1_gelu_erf(float v) {
    const float x = v * 0.70710678118654752440f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = poly * t * __expf(-ax * ax);
    return 0.5f * v * (x >= 0.f ? 2.0f - e : e);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- optional per-kernel timing (fo1_profile_*) ---------------------------------------
// When enabled every FO1_LAUNCH goes through hipExtLaunchKernelGGL with a start/stop event pair, which the runtime fills
// from the dispatch packet's own begin/end timestamps (the clocks rocprofv3's kernel trace reads), so a row's time is
// kernel execution only -- no inter-packet latency.  Disabled (the default, and always during throughput timing and
// graph capture) it is a plain launch.
bool profile_enabled();
void profile_events(const char* name, double work, hipEvent_t* e0, hipEvent_t* e1);

#define FO1_LAUNCH(name, work, kernel, grid, block, shmem, st, ...)                                   \
    do {                                                                                              \
        if (fo1::profile_enabled()) {                                                                 \
            hipEvent_t _e0, _e1;                                                                      \
            fo1::profile_events(name, (double)(work), &_e0, &_e1);                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, _e0, _e1, 0, __VA_ARGS__);          \
        } else {                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                          \
        }                                                                                             \
        FO1_CHECK_LAUNCH();                                                                           \
    } while (0)

}  // namespace fo1
