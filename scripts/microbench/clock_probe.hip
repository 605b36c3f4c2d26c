// What clock does a CU run at under load?  Each workgroup (one per CU, 8 waves) spins for ~200 us on (a) scalar adds only, (b) dense
// v_mfma_f32_32x32x16_bf16 on random operands, (c) the same on zero operands, and reports shader cycles (s_memtime) / wall time
// (s_memrealtime, 100 MHz) = the effective clock, plus the MFMA rate it implies.  Calibrates the roofline peak of bench.py: 2.5 PFLOP/s
// dense bf16 assumes 2.4 GHz.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe scripts/microbench/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ seed, int iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    union { unsigned u[4]; bf16x8 v; } a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned s = MODE == 2 ? 0u : seed[(threadIdx.x * 4 + i) & 4095];
        a.u[i] = (s & 0x807F807Fu) | 0x3F003F00u;                     // bf16 pairs in [0.5, 1) with random signs / mantissas
        b.u[i] = ((s >> 3) & 0x807F807Fu) | 0x3F003F00u;
        if (MODE == 2) a.u[i] = b.u[i] = 0;
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    unsigned long long c0 = 0, t0 = 0;
    int x = (int)blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); t0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(x));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
    if (s == 123.456f || x == -7) sink[0] = s;
}

template <int MODE>
static int run(const char* name, const unsigned* seed, unsigned long long* out, float* sink, int iters) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, seed, iters, out, sink);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz, us;
    for (int i = 0; i < 256; ++i) { ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); us.push_back(h[2 * i + 1] / 100.0); }
    std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
    const double mfma_per_wave = (double)iters * 32, cyc = ghz[128] * us[128] * 1e3;
    printf("{\"load\": \"%s\", \"median_clock_ghz\": %.3f, \"min_clock_ghz\": %.3f, \"max_clock_ghz\": %.3f, \"median_us\": %.1f", name, ghz[128], ghz[0], ghz[255], us[128]);
    if (MODE) printf(", \"cycles_per_mfma_per_simd\": %.1f, \"chip_tflops\": %.0f", cyc / (mfma_per_wave * 2), 256.0 * 8 * mfma_per_wave * 32768.0 / (us[128] * 1e-6) / 1e12);
    printf("}\n");
    return 0;
}

int main() {
    unsigned* seed; unsigned long long* out; float* sink;
    CK(hipMalloc(&seed, 4096 * 4)); CK(hipMalloc(&out, 512 * 8)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned> h(4096);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    CK(hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    if (run<0>("scalar adds only", seed, out, sink, 4000)) return 1;
    if (run<1>("dense mfma 32x32x16 bf16, random operands", seed, out, sink, 400)) return 1;
    if (run<1>("dense mfma, random operands, 10x longer", seed, out, sink, 4000)) return 1;
    if (run<2>("dense mfma, zero operands", seed, out, sink, 4000)) return 1;
    if (run<0>("scalar adds only (again)", seed, out, sink, 4000)) return 1;
    return 0;
}
