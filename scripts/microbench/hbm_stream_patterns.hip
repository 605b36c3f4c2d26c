// How fast does a weight tile stream from HBM into a CU's LDS ring, by layout?  Each workgroup (4 waves, like gemm_bt_ring_kernel<128,128,3>'s W
// half) walks weight tiles of 128 rows x 2048 bf16 columns in K tiles of 64 columns (16 KB per K tile, 16 LDS-DMA pieces, 2 K tiles in flight,
// one barrier per K tile) from a 2 GiB buffer (nothing stays in the 256 MiB Infinity Cache):
//   rows:   the row-major [N][K] layout of the checkpoint — a K tile = 128 pieces of 128 B at a 4 KiB stride
//   tiled:  a copy pre-tiled as [n tile][k tile][128 rows][128 B] — a K tile = one contiguous 16 KiB block
// for 172 workgroups (the decode pool's gate/up at 128 rows) and 256 / 512 (every CU busy / two per CU).  Prints TB/s.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/hsp scripts/microbench/hbm_stream_patterns.hip && /tmp/hsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NK = 32;                       // K tiles per weight tile (K = 2048)
constexpr size_t TILE_BYTES = 128 * 4096;    // one weight tile: 128 rows x 4 KiB

template <int TILED>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ base, int tiles_total, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 stages x 16 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    // lane's offset inside a K tile for its 4 pieces (piece = wave * 4 + i: 8 rows x 128 B)
    uint32_t off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i, row = piece * 8 + (lane >> 3), chunk = lane & 7;
        off[i] = TILED ? (uint32_t)(piece * 1024 + lane * 16) : (uint32_t)(row * 4096 + chunk * 16);
    }
    int stage = 0;
    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        const char* tb = base + (size_t)tile * TILE_BYTES;
        auto issue = [&](int kt, int st) {
            const char* ub = tb + (TILED ? (size_t)kt * 16384 : (size_t)kt * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t dst = lds0 + st * 16384 + (wave * 4 + i) * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[i]), "s"(ub), "s"(dst) : "memory");
            }
        };
        issue(0, stage);
        issue(1, (stage + 1) % 3);
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 1 < NK) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < NK) issue(kt + 2, (stage + 2) % 3);
            stage = (stage + 1) % 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (*reinterpret_cast<const unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

template <int TILED>
static int run(const char* name, const char* buf, unsigned* sink, int grid, int tiles_total, const char* warm = nullptr) {
    CK(hipFuncSetAttribute((const void*)stream_kernel<TILED>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_kernel<TILED>), dim3(grid), dim3(256), 49152, 0, warm ? warm : buf, tiles_total, sink);      // warm-up (elsewhere for the one-round runs)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<TILED>), dim3(grid), dim3(256), 49152, 0, buf, tiles_total, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)tiles_total * TILE_BYTES;
    printf("{\"layout\": \"%s\", \"workgroups\": %d, \"weight_tiles\": %d, \"us\": %.1f, \"tb_per_s\": %.2f, \"gb_per_s_per_workgroup\": %.1f, \"us_per_k_tile\": %.3f}\n", name, grid,
           tiles_total, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / grid / (ms * 1e-3) / 1e9, ms * 1e3 / ((double)(tiles_total + grid - 1) / grid * NK));
    return 0;
}

int main() {
    char* buf; unsigned* sink;
    const int tiles_total = 4096;                 // 2 GiB
    CK(hipMalloc(&buf, (size_t)tiles_total * TILE_BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, (size_t)tiles_total * TILE_BYTES));
    for (int grid : {172, 256, 512}) {
        if (run<0>("rows", buf, sink, grid, tiles_total)) return 1;
        if (run<1>("tiled", buf, sink, grid, tiles_total)) return 1;
    }
    // one round only, like the pool's gate/up (172 tiles, each workgroup exactly one): launch + ramp included
    const char* far = buf + (size_t)3072 * TILE_BYTES;
    if (run<0>("rows, one tile per workgroup", buf, sink, 172, 172, far)) return 1;
    if (run<1>("tiled, one tile per workgroup", buf + (size_t)512 * TILE_BYTES, sink, 172, 172, far)) return 1;
    if (run<0>("rows, one tile per workgroup", buf + (size_t)1024 * TILE_BYTES, sink, 256, 256, far)) return 1;
    if (run<1>("tiled, one tile per workgroup", buf + (size_t)2048 * TILE_BYTES, sink, 256, 256, far)) return 1;
    return 0;
}
