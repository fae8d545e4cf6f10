// dpp_probe.hip -- semantics of the cross-lane primitives the wave-cooperative kernel relies on (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CTRL, int BANK>
__device__ __forceinline__ int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false); }

__global__ void k(int *out) {
    const int l = threadIdx.x;
    int v = l;
    out[0 * 64 + l] = dpp<0x121, 0xF>(-1, v);       // row_ror:1
    out[1 * 64 + l] = dpp<0x123, 0xF>(-1, v);       // row_ror:3
    out[2 * 64 + l] = dpp<0x128, 0x3>(-1, v);       // row_ror:8, banks 0-1 only
    out[3 * 64 + l] = dpp<0x128, 0xC>(-1, v);       // row_ror:8, banks 2-3 only
    out[4 * 64 + l] = dpp<0x111, 0xF>(-1, v);       // row_shr:1
    out[5 * 64 + l] = dpp<0x101, 0xF>(-1, v);       // row_shl:1
    out[6 * 64 + l] = __builtin_amdgcn_ds_bpermute(((l & 48) + 5) * 4, v);   // broadcast lane 5 of my row
    out[7 * 64 + l] = __builtin_amdgcn_ds_bpermute((((l >> 4) ^ 1) * 16 + (l & 15)) * 4, v);   // swap rows 0<->1, 2<->3
}
int main() {
    int *d, h[8 * 64];
    CHECK(hipMalloc(&d, sizeof h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    const char *names[8] = {"row_ror:1", "row_ror:3", "row_ror:8 bank0-1", "row_ror:8 bank2-3", "row_shr:1", "row_shl:1", "bperm bcast5", "bperm rowswap"};
    for (int t = 0; t < 8; t++) {
        printf("%-18s", names[t]);
        for (int l = 0; l < 20; l++) printf(" %3d", h[t * 64 + l]);
        printf(" ...");
        for (int l = 44; l < 52; l++) printf(" %3d", h[t * 64 + l]);
        printf("\n");
    }
    return 0;
}
