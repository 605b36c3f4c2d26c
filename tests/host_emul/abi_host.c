/* A plain C99 host of the C-ABI (no Python, no torch): loads libfo1hip.so, reads the ABI version, asks for a workspace size and
 * provokes an argument error — the calls a non-Python integrator makes first.  No device work: runs without a GPU.
 * Built and run by tests/test_abi.py. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "fo1.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    int (*abi)(void) = (int (*)(void))dlsym(h, "fo1_abi_version");
    const char* (*last_error)(void) = (const char* (*)(void))dlsym(h, "fo1_last_error");
    size_t (*ws_bytes)(const fo1_hfre_source_t*, int, int) = (size_t (*)(const fo1_hfre_source_t*, int, int))dlsym(h, "fo1_hfre_workspace_bytes");
    int (*pool)(const fo1_hfre_source_t*, int, const float*, int, const float*, float, float, int, int, float, float, float*, int, int, void*,
                size_t, void*) = (int (*)(const fo1_hfre_source_t*, int, const float*, int, const float*, float, float, int, int, float, float,
                                          float*, int, int, void*, size_t, void*))dlsym(h, "fo1_hfre_region_pool");
    if (!abi || !last_error || !ws_bytes || !pool) { fprintf(stderr, "missing symbol\n"); return 4; }
    fo1_hfre_source_t src;
    memset(&src, 0, sizeof src);
    src.data = (const void*)0x1000;      /* never dereferenced on the host */
    src.H = 120; src.W = 160; src.C = 256; src.ld = 256; src.roi_H = 120; src.roi_W = 160;
    src.spatial_scale = 0.25f; src.box_space = 0; src.out_offset = 0;
    const size_t need = ws_bytes(&src, 1, 100);
    /* NULL boxes: must come back as an argument error with a message, not a crash */
    const int rc = pool(&src, 1, (const float*)0, 100, (const float*)0, 1.f, 1.f, 7, 0, 1.f, 1.f, (float*)0, 256, 256, (void*)0, 0, (void*)0);
    printf("abi=%d workspace=%zu rc=%d error=%s\n", abi(), need, rc, last_error());
    return (abi() > 0 && need > 0 && rc < 0 && strlen(last_error()) > 0) ? 0 : 5;
}
