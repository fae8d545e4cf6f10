// stepbench.hip -- A/B of the Montgomery ladder STEP on the whole chip: the library's unsigned 28-bit limbs
// (gf28.hpp, montgomery.hpp ml_step_sel) against the signed, register-paired limbs of gf28s.hpp (ml_step_sel_s).
// Each lane walks N steps from the same start with the same swap bits; the two final states are compared
// limb by limb in canonical form (every lane), then both loops are timed at two waves per SIMD.
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -DGD_NO_PAIRED_ADDS=1 -o tools/stepbench tools/stepbench.hip
// (GD_NO_PAIRED_ADDS: the reference side is round 4's step as it was; with gf28.hpp's pair-wise additions the UNSIGNED step
// measures 17 % slower -- 14 899 cycles -- which is one more reason the ladders' unit is built without them)
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include "montgomery.hpp"

using namespace gd;

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

__device__ __forceinline__ fe load_fe(const uint32_t *p) {
    fe x;
#pragma unroll
    for (int i = 0; i < 16; i++) x.v[i] = p[i] & M28;
    return x;
}

// out: 4 canonical field elements per lane (x2, z2, x3, z3 after n steps)
template <int SIGNED>
__global__ void __launch_bounds__(256, 2) k_steps(uint32_t *out, const uint32_t *in, int n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const fe x1 = load_fe(in + t * 64), a = load_fe(in + t * 64 + 16), b = load_fe(in + t * 64 + 32);
    uint32_t bits = in[t * 64 + 48];
    fe r[4];
    if (SIGNED) {
        MlStateS st;
        st.x2 = sfe_from_fe(a);
        st.z2 = sfe_from_fe(b);
        st.x3 = sfe_from_fe(x1);
        st.z3 = sfe_from_fe(a);
        const smultiplier m1 = s_multiplier(sfe_from_fe(x1));
#pragma unroll 1
        for (int k = 0; k < n; k++) {
            const bool sw = bits & 1;
            bits = (bits >> 1) | (bits << 31);
            ml_step_sel_s(st, m1, sw);
        }
        r[0] = sfe_to_fe(st.x2); r[1] = sfe_to_fe(st.z2); r[2] = sfe_to_fe(st.x3); r[3] = sfe_to_fe(st.z3);
    } else {
        fe x2 = a, z2 = b, x3 = x1, z3 = a;
#pragma unroll 1
        for (int k = 0; k < n; k++) {
            const bool sw = bits & 1;
            bits = (bits >> 1) | (bits << 31);
            ml_step_sel(x2, z2, x3, z3, x1, sw);
        }
        r[0] = x2; r[1] = z2; r[2] = x3; r[3] = z3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const fe c = fe_strong(r[j]);
#pragma unroll
        for (int i = 0; i < 16; i++) out[t * 64 + j * 16 + i] = c.v[i];
    }
}

template <int SIGNED>
static double run(const char *name, uint32_t *d_out, const uint32_t *d_in, int n) {
    const int blocks = 256 * 2;   // 256 CUs, 4 waves per block: two waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_steps<SIGNED>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, 8);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_steps<SIGNED>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / n / 2;
    printf("%-10s %8.3f ms for %d steps  %8.1f SIMD-cycles per wave-step at 2 waves/SIMD (nominal 2.4 GHz)\n", name, best, n,
           cyc);
    return cyc;
}

int main() {
    const size_t lanes = 256 * 2 * 256;
    const size_t words = lanes * 64;
    uint32_t *h = (uint32_t *)malloc(words * 4), *ha = (uint32_t *)malloc(words * 4), *hb = (uint32_t *)malloc(words * 4);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < words; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (uint32_t)(s >> 11);
    }
    uint32_t *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, words * 4));
    CHECK(hipMalloc(&d_a, words * 4));
    CHECK(hipMalloc(&d_b, words * 4));
    CHECK(hipMemcpy(d_in, h, words * 4, hipMemcpyHostToDevice));
    // parity of the two step functions: 446 steps, every lane
    hipLaunchKernelGGL(k_steps<0>, dim3(512), dim3(256), 0, 0, d_a, d_in, 446);
    hipLaunchKernelGGL(k_steps<1>, dim3(512), dim3(256), 0, 0, d_b, d_in, 446);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ha, d_a, words * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb, d_b, words * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < words; i++) bad += ha[i] != hb[i];
    printf("parity after 446 steps, %zu lanes x 4 field elements (canonical limbs): %zu mismatching words\n", lanes, bad);
    const double c0 = run<0>("unsigned", d_a, d_in, 446);
    const double c1 = run<1>("signed", d_b, d_in, 446);
    printf("signed / unsigned = %.4f\n", c1 / c0);
    return bad != 0;
}
