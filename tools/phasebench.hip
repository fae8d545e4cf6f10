// phasebench.hip -- do two waves on one SIMD co-issue their simple VALU ops only when they are in
// phase?  One 512-thread workgroup per CU puts two waves of the SAME workgroup on every SIMD, so an
// s_barrier per loop iteration can re-align them; the loop body is a run of R MACs followed by a run
// of R simple ops (the shape of the field arithmetic).  Compared: no barrier / barrier per iteration.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/phasebench tools/phasebench.hip && ./tools/phasebench
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

#define M0 "v_mad_u64_u32 %0, vcc, %8, %9, %0\n"
#define M1 "v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
#define M2 "v_mad_u64_u32 %2, vcc, %8, %9, %2\n"
#define M3 "v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
#define S0 "v_add_u32_e32 %4, %8, %4\n"
#define S1 "v_and_b32_e32 %5, %9, %5\n"
#define S2 "v_add_u32_e32 %6, %8, %6\n"
#define S3 "v_lshrrev_b32_e32 %7, 1, %7\n"
#define STR2(x) #x
#define STR(x) STR2(x)

constexpr int ITERS = 4096;

// REP4 = R/4: the body holds R MACs then R simple ops
template <int SYNC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_runs(uint64_t *out, uint32_t a0, uint32_t b0) {
    uint64_t m0, m1, m2, m3;
    uint32_t s0, s1, s2, s3;
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    m0 = a; m1 = a * 2; m2 = a * 3; m3 = a * 4; s0 = b; s1 = b * 3; s2 = b * 5; s3 = b * 7;
    for (int it = 0; it < ITERS; it++) {
        asm volatile(".rept " STR(REP4) "\n" M0 M1 M2 M3 "\n.endr\n.rept " STR(REP4) "\n" S0 S1 S2 S3 "\n.endr"
                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                     : "v"(a), "v"(b) : "vcc");
        if (SYNC) __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = m0 ^ m1 ^ m2 ^ m3 ^ s0 ^ s1 ^ s2 ^ s3;
}

template <int SYNC, int THREADS>
static void run(const char *name, uint64_t *out, int cus, int blocks_per_cu) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * blocks_per_cu;
    hipLaunchKernelGGL((k_runs<SYNC, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, 1u, 2u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_runs<SYNC, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, 1u, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts = (double)ITERS * REP4 * 8;             // per wave
    const int waves_per_simd = THREADS / 256 * blocks_per_cu;  // 4 SIMDs per CU
    printf("%-34s runs of %3d: %8.3f ms  %.2f cycles per instruction per SIMD (nominal 2.4 GHz)\n", name, REP4 * 4,
           best, best * 1e-3 * 2.4e9 / insts / waves_per_simd);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint64_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 2 * 512 * 8));
    run<0, 256>("2 x 256-thread blocks per CU", out, cus, 2);
    run<0, 512>("1 x 512-thread block, no barrier", out, cus, 1);
    run<1, 512>("1 x 512-thread block, s_barrier", out, cus, 1);
    return 0;
}
