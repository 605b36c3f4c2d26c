// Per-CU fetch throughput of the three ways a GEMM tile can reach a CU on gfx950, from L2-resident data (each workgroup re-reads its own
// 64 KiB window): (a) global_load_dwordx4 into VGPRs, (b) global_load_lds_dwordx4 (LDS-DMA, saddr form) into an LDS ring, (c) both at once,
// half the bytes each.  One workgroup per CU (grid 256), 256 or 512 threads.  Prints GB/s per CU and TB/s per chip.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_fetch scripts/microbench/cu_fetch_paths.hip && /tmp/cu_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: VGPR loads only; 1: LDS-DMA only; 2: half / half.  `iters` sweeps of the 64 KiB window per workgroup.
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void fetch_kernel(const char* __restrict__ base, int iters, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KiB ring for the DMA path
    constexpr int WAVES = THREADS / 64;
    constexpr int PIECES = 64;                                        // 1 KiB pieces per sweep
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* win = base + (size_t)blockIdx.x * 65536;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 vals[PIECES / WAVES];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < PIECES / WAVES; ++p) {
            const int piece = p * WAVES + wave;
            const bool dma = MODE == 1 || (MODE == 2 && (p & 1));
            if (dma) {
                const uint32_t dst = lds0 + piece * 1024;
                const uint32_t off = piece * 1024 + lane * 16;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(win), "s"(dst) : "memory");
            } else {
                u32x4 v;        // (asm: a plain C++ load of a loop-invariant address is hoisted out of the sweep loop)
                const uint32_t off = piece * 1024 + lane * 16;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(win) : "memory");
                vals[p] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE != 1) {
#pragma unroll
            for (int p = 0; p < PIECES / WAVES; ++p)
                if (MODE == 0 || !(p & 1)) acc ^= vals[p];
        }
    }
    if (MODE != 0) acc.x ^= *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 4);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int THREADS>
static int run(const char* name, const char* buf, unsigned* sink, int grid) {
    const int iters = 400;
    CK(hipFuncSetAttribute((const void*)fetch_kernel<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fetch_kernel<MODE, THREADS>), dim3(grid), dim3(THREADS), 65536, 0, buf, 20, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((fetch_kernel<MODE, THREADS>), dim3(grid), dim3(THREADS), 65536, 0, buf, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_wg = 65536.0 * iters;
    printf("{\"path\": \"%s\", \"threads\": %d, \"workgroups\": %d, \"us\": %.1f, \"gb_per_s_per_cu\": %.1f, \"tb_per_s_chip\": %.2f}\n", name, THREADS, grid, ms * 1e3,
           bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg * grid / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    char* buf; unsigned* sink;
    const size_t n = (size_t)256 * 65536;
    CK(hipMalloc(&buf, n)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned> h(n / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(buf, h.data(), n, hipMemcpyHostToDevice));
    for (int grid : {256, 32}) {
        if (run<0, 256>("vgpr", buf, sink, grid)) return 1;
        if (run<1, 256>("lds_dma", buf, sink, grid)) return 1;
        if (run<2, 256>("half_half", buf, sink, grid)) return 1;
        if (run<0, 512>("vgpr", buf, sink, grid)) return 1;
        if (run<1, 512>("lds_dma", buf, sink, grid)) return 1;
        if (run<2, 512>("half_half", buf, sink, grid)) return 1;
    }
    return 0;
}
