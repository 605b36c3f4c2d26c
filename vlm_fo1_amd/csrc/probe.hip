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


// ---- HBM counter calibration (VERDICT r5 #2a): kernels that move a KNOWN number of bytes in the access patterns of the kernels whose
// rocprofv3 FETCH_SIZE / WRITE_SIZE readings the roofline quotes.  MI355X_MICROARCH.md: FETCH_SIZE reads half of a wide coalesced stream,
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
//   mode 0  stream store: every lane 16 B, a wave = 1 KB contiguous (the plain epilogues, norms, copies)
//   mode 1  tile store: the 256 x 256 GEMM's coalesced epilogue — a wave instruction = 8 row segments of 128 B at the row pitch `ld` bytes,
//           a workgroup of 512 threads writes a 256-row x 512-byte tile in 16 sweeps, tiles walk the [rows, ld] matrix
//   mode 2  stream load: every lane 16 B contiguous (what the guide calibrated: x2)
//   mode 3  tile load by LDS-DMA: global_load_lds_dwordx4, a wave instruction = 8 rows x 128 B at the row pitch (the GEMM's A / W staging)
//   mode 4  stream store, 8 B per lane;  mode 5: stream store, 4 B per lane (the fragment-shaped epilogues)
typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void traffic_probe_kernel(int mode, unsigned char* __restrict__ buf, long long bytes, long long ld, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[512 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const probe_u32x4 val = {0x3F803F80u + (unsigned)tid, 0x40004000u, 0xBF80BF80u, (unsigned)blockIdx.x};
    probe_u32x4 acc = {0u, 0u, 0u, 0u};
    if (mode == 0 || mode == 2 || mode == 4 || mode == 5) {
        const int w = mode == 4 ? 8 : (mode == 5 ? 4 : 16);
        const long long n = bytes / w;
        for (long long i = (long long)blockIdx.x * 512 + tid; i < n; i += (long long)gridDim.x * 512) {
            if (mode == 0) *reinterpret_cast<probe_u32x4*>(buf + i * 16) = val;
            else if (mode == 4) *reinterpret_cast<uint2*>(buf + i * 8) = uint2{val.x, val.y};
            else if (mode == 5) *reinterpret_cast<unsigned*>(buf + i * 4) = val.x;
            else { const probe_u32x4 v = *reinterpret_cast<const probe_u32x4*>(buf + i * 16); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        }
    } else {
        // [rows, ld bytes] matrix cut into tiles of 256 rows x 512 bytes
        const long long rows = bytes / ld, tiles_n = ld / 512, tiles = (rows / 256) * tiles_n;
        for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
            const long long r0 = (t / tiles_n) * 256, c0 = (t % tiles_n) * 512;
            if (mode == 1) {
                // lane -> (row of an 8-row group, 16-byte piece of a 128-byte segment); a wave covers 8 rows x 128 B per store, 4 stores per 512-byte row
#pragma unroll
                for (int sw = 0; sw < 16; ++sw) {
                    const int r = (sw >> 2) * 64 + wave * 8 + (lane >> 3), seg = sw & 3;
                    *reinterpret_cast<probe_u32x4*>(buf + (r0 + r) * ld + c0 + seg * 128 + (lane & 7) * 16) = val;
                }
            } else {
#pragma unroll
                for (int sw = 0; sw < 16; ++sw) {
                    const int r = (sw >> 2) * 64 + wave * 8 + (lane >> 3), seg = sw & 3;
                    const unsigned char* src = buf + (r0 + r) * ld + c0 + seg * 128 + (lane & 7) * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const probe_u32x4 v = *reinterpret_cast<const probe_u32x4*>(lds + tid * 16);
                acc.x ^= v.x; acc.y ^= v.y;
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1.f;     // keeps the loads alive
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


// Moves exactly `bytes` (a multiple of 256 rows x ld for the tile modes; ld a multiple of 512) in one of the access patterns above: run it
// under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide (scripts/pmc_calibrate.py).  buf: device, >= bytes.
int fo1_traffic_probe(int mode, void* buf, long long bytes, long long ld, int workgroups, void* sink, void* stream) {
    using namespace fo1;
    FO1_CHECK_ARG(buf && sink && mode >= 0 && mode <= 5 && bytes > 0 && bytes % 16 == 0, "traffic_probe: bad arguments");
    if (mode == 1 || mode == 3) FO1_CHECK_ARG(ld >= 512 && ld % 512 == 0 && bytes % (256 * ld) == 0, "traffic_probe: tile modes need ld %% 512 == 0 and bytes %% (256 ld) == 0");
    static const char* names[6] = {"traffic_probe_stream_store16", "traffic_probe_tile_store", "traffic_probe_stream_load16", "traffic_probe_tile_load_lds",
                                   "traffic_probe_stream_store8", "traffic_probe_stream_store4"};
    FO1_LAUNCH(names[mode], (double)bytes, traffic_probe_kernel, dim3(workgroups > 0 ? workgroups : 2048), dim3(512), 0, (hipStream_t)stream, mode, (unsigned char*)buf, bytes, ld, (float*)sink);
    return FO1_OK;
}

}  // extern "C"

#endif   // FO1_ENABLE_AB
