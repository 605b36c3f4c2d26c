// Instrumentation (include/fo1_ab.h: test / bench build only — bench.py's `roofline.sustained_mfma` leg loads it after the timed region): what clock the matrix pipes sustain on THIS box.  The dense bf16 peak a
// roofline is priced against (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x 1024 flop/cycle at 2.4 GHz; under load the chip clocks to its power
// budget (DVFS), so the attainable rate is peak x (sustained clock / 2.4 GHz) even for a loop that issues an MFMA every cycle it can.
// fo1_mfma_clock_probe runs such a loop — register-resident operands, no memory traffic, 8 waves on every CU — and returns per workgroup
// the shader cycles (s_memtime) and the 100 MHz wall ticks (s_memrealtime) it took.  bench.py reports it next to the roofline
// (`roofline.sustained_mfma`): the ceiling of ANY bf16 MFMA kernel on this box and data, measured in the same process.
#include "common.h"

#ifdef FO1_ENABLE_AB

namespace fo1 {

typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));

// operands 0: zeros (the least switching activity), 1: pseudo-random bf16 in +-[0.5, 1)
__global__ __launch_bounds__(512) void mfma_clock_probe_kernel(int operands, int iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    union { unsigned u[4]; probe_bf16x8 v; } a, b;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s = s * 1664525u + 1013904223u;
        a.u[i] = operands ? ((s & 0x807F807Fu) | 0x3F003F00u) : 0u;
        b.u[i] = operands ? (((s >> 3) & 0x807F807Fu) | 0x3F003F00u) : 0u;
    }
    probe_f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    unsigned long long c0 = 0, t0 = 0;
    __syncthreads();
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); t0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[j], 0, 0, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][9];
    if (t == 123.456f) sink[0] = t;       // keeps the accumulators alive
}

}  // namespace fo1

extern "C" {

// out: device, uint64 [workgroups][2] = {shader cycles, 100 MHz ticks} of `iters` x 32 MFMAs (v_mfma_f32_32x32x16_bf16) per wave, 8 waves per
// workgroup; sink: device float (never written in practice).  flop of the launch = workgroups x 8 x iters x 32 x 32768.
int fo1_mfma_clock_probe(int operands, int iters, int workgroups, void* out, void* sink, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(out && sink && iters >= 1 && workgroups >= 1 && (operands == 0 || operands == 1), "mfma_clock_probe: bad arguments");
    FO1_LAUNCH("mfma_clock_probe", (double)workgroups * 8.0 * iters * 32.0 * 32768.0, mfma_clock_probe_kernel, dim3(workgroups), dim3(512), 0,
               (hipStream_t)stream, operands, iters, (unsigned long long*)out, (float*)sink);
    return FO1_OK;
}

}  // extern "C"

#endif   // FO1_ENABLE_AB
