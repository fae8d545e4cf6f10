// ubench.hip -- instruction-rate microbenchmarks for the integer-multiply roofline on gfx950.
//
// The Ed448 ladder is VALU-bound on 32x32->64 multiply-accumulates, not HBM- or MFMA-bound
// (SURVEY.md section 8d), so the honest ceiling is the measured issue rate of
// v_mad_u64_u32 and of the carry-handling instructions around it.  This tool measures,
// per instruction, SIMD cycles per wave-instruction at 1/2/4 waves per SIMD with 1 or 8
// independent dependency chains, from s_memtime and from wall clock.
//
//   hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 32;  // instructions per loop iteration

// Each kernel: K independent chains, UNROLL instructions per iteration round-robin over chains.
#define DEF_KERNEL64(NAME, ASM)                                                                    \
    template <int K>                                                                               \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t a0, uint32_t b0,           \
                                                unsigned long long *cyc) {                         \
        uint64_t acc[8];                                                                           \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
        for (int i = 0; i < 8; i++) acc[i] = a * (i + 1);                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITERS; it++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                   \
                asm volatile(ASM : "+v"(acc[u % K]) : "v"(a), "v"(b) : "vcc");                     \
            }                                                                                      \
        }                                                                                          \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        uint64_t r = 0;                                                                            \
        for (int i = 0; i < 8; i++) r ^= acc[i];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                            \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
    }

#define DEF_KERNEL32(NAME, ASM)                                                                    \
    template <int K>                                                                               \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t a0, uint32_t b0,           \
                                                unsigned long long *cyc) {                         \
        uint32_t acc[8];                                                                           \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
        for (int i = 0; i < 8; i++) acc[i] = a * (i + 1);                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITERS; it++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                   \
                asm volatile(ASM : "+v"(acc[u % K]) : "v"(a), "v"(b) : "vcc");                     \
            }                                                                                      \
        }                                                                                          \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        uint64_t r = 0;                                                                            \
        for (int i = 0; i < 8; i++) r ^= acc[i];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                            \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
    }

DEF_KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
DEF_KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
DEF_KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
DEF_KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
DEF_KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
DEF_KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
DEF_KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL32(k_mul_u32_u24, "v_mul_u32_u24_e32 %0, %1, %0")
DEF_KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24_e32 %0, %1, %0")
DEF_KERNEL32(k_add_u32, "v_add_u32_e32 %0, %1, %0")
DEF_KERNEL32(k_add3_u32, "v_add3_u32 %0, %1, %2, %0")
DEF_KERNEL32(k_and_b32, "v_and_b32_e32 %0, %1, %0")
DEF_KERNEL32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL32(k_lshrrev_b32, "v_lshrrev_b32_e32 %0, 1, %0")
DEF_KERNEL32(k_alignbit_b32, "v_alignbit_b32 %0, %1, %0, 28")
DEF_KERNEL32(k_add_co_u32, "v_add_co_u32_e32 %0, vcc, %1, %0")
DEF_KERNEL32(k_addc_co_u32, "v_addc_co_u32_e32 %0, vcc, %1, %0, vcc")
DEF_KERNEL32(k_cndmask_b32, "v_cndmask_b32_e32 %0, %1, %0, vcc")
DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %0, %0")
DEF_KERNEL32(k_mad_u16, "v_mad_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 4, 28")
DEF_KERNEL32(k_mov_b32, "v_mov_b32_e32 %0, %1")

typedef void (*kern_t)(uint64_t *, uint32_t, uint32_t, unsigned long long *);

struct Entry {
    const char *name;
    kern_t k1, k8;
};
#define ENT(n) {#n, n<1>, n<8>}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    Entry ents[] = {ENT(k_mad_u64_u32), ENT(k_mad_i64_i32), ENT(k_mul_lo_u32), ENT(k_mul_hi_u32),
                    ENT(k_mad_u32_u24), ENT(k_mul_u32_u24), ENT(k_mul_hi_u32_u24), ENT(k_mad_u16),
                    ENT(k_dot4_u32_u8), ENT(k_add_u32), ENT(k_add3_u32), ENT(k_and_b32), ENT(k_and_or_b32),
                    ENT(k_bfe_u32), ENT(k_lshrrev_b32), ENT(k_alignbit_b32), ENT(k_add_co_u32),
                    ENT(k_addc_co_u32), ENT(k_cndmask_b32), ENT(k_mov_b32), ENT(k_lshl_add_u64),
                    ENT(k_lshrrev_b64), ENT(k_fma_f32), ENT(k_pk_fma_f32), ENT(k_fma_f64)};
    uint64_t *out;
    unsigned long long *cyc;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * 8));
    CHECK(hipMalloc(&cyc, (size_t)max_blocks * 8));
    unsigned long long *hcyc = (unsigned long long *)malloc((size_t)max_blocks * 8);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-18s %5s %6s | %10s %10s | %12s\n", "instr", "w/SIMD", "chains", "cyc/inst", "cyc(wall)", "Ginst/s chip");
    const char *only = argc > 1 ? argv[1] : nullptr;
    for (auto &e : ents) {
        if (only && !strstr(e.name, only)) continue;
        for (int wps : {1, 2, 4, 8}) {
            for (int chains : {1, 8}) {
                kern_t k = chains == 1 ? e.k1 : e.k8;
                int blocks = cus * wps;  // 256-thread blocks: wps blocks/CU -> wps waves per SIMD
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 6789u, cyc);  // warm-up
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                const int reps = 4;
                for (int r = 0; r < reps; r++)
                    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 6789u, cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(hcyc, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
                double avg = 0;
                for (int b = 0; b < blocks; b++) avg += (double)hcyc[b];
                avg /= blocks;
                double insts_per_wave = (double)ITERS * UNROLL;
                // a SIMD hosts `wps` waves; cycles the SIMD spends per wave-instruction:
                double cyc_per_inst = avg / insts_per_wave / wps;
                double total_wave_insts = insts_per_wave * blocks * 4.0 * reps;
                double ginst = total_wave_insts / (ms * 1e-3) / 1e9;  // wave-instructions per second
                // wall-clock cycles per wave-instruction per SIMD at 2.4 GHz nominal
                double cyc_wall = (ms * 1e-3 / reps) * 2.4e9 / insts_per_wave / wps;
                printf("%-18s %5d %6d | %10.2f %10.2f | %12.1f\n", e.name + 2, wps, chains, cyc_per_inst, cyc_wall,
                       ginst);
            }
        }
    }
    return 0;
}
