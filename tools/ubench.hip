// ubench.hip -- instruction-rate microbenchmarks for the integer-multiply roofline on gfx950.
//
// The Ed448 ladder is VALU-bound on 32x32->64 multiply-accumulates, not HBM- or MFMA-bound
// (SURVEY.md section 8d), so the honest ceiling is the measured issue rate of
// v_mad_u64_u32 and of the carry-handling instructions around it.  This tool measures,
// per instruction, SIMD cycles per wave-instruction at 1/2/4/8 waves per SIMD with 1 or 8
// independent dependency chains, from wall clock (nominal 2.4 GHz) -- plus three instruction
// mixes shaped like the ladder's inner loops.
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
constexpr int UNROLL = 32;  // instructions per loop iteration (REPT x 8)

// Each kernel: one asm statement per loop iteration (hipcc pads every asm statement with an
// s_nop, which would steal issue slots), containing REPT x 8 instructions on 8 independent
// registers (K = 8) or on one register (K = 1, a dependent chain).
#define REPT 4  // 4 x 8 = 32 instructions per iteration
#define STR2(x) #x
#define STR(x) STR2(x)

#define DEF_KERNEL(NAME, TYPE, I0, I1, I2, I3, I4, I5, I6, I7)                                     \
    template <int K>                                                                               \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t a0, uint32_t b0,           \
                                                unsigned long long *cyc) {                         \
        TYPE r0, r1, r2, r3, r4, r5, r6, r7;                                                       \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
        r0 = a; r1 = a * 2; r2 = a * 3; r3 = a * 4; r4 = a * 5; r5 = a * 6; r6 = a * 7; r7 = a * 8; \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITERS; it++) {                                                       \
            if (K == 8)                                                                            \
                asm volatile(".rept " STR(REPT) "\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n.endr" \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(a), "v"(b) : "vcc");                                            \
            else                                                                                   \
                asm volatile(".rept " STR(REPT) "\n" I0 "\n" I0 "\n" I0 "\n" I0 "\n" I0 "\n" I0 "\n" I0 "\n" I0 "\n.endr" \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(a), "v"(b) : "vcc");                                            \
        }                                                                                          \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        uint64_t r = (uint64_t)r0 ^ (uint64_t)r1 ^ (uint64_t)r2 ^ (uint64_t)r3 ^ (uint64_t)r4 ^ (uint64_t)r5 ^ \
                     (uint64_t)r6 ^ (uint64_t)r7;                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                            \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
    }
// same instruction on 8 registers; operands: %0..%7 accumulators, %8 = a, %9 = b
#define ALL8(NAME, TYPE, PRE, POST)                                                                \
    DEF_KERNEL(NAME, TYPE, PRE "%0" POST("%0"), PRE "%1" POST("%1"), PRE "%2" POST("%2"), PRE "%3" POST("%3"), \
               PRE "%4" POST("%4"), PRE "%5" POST("%5"), PRE "%6" POST("%6"), PRE "%7" POST("%7"))

#define P_MAD(r) ", vcc, %8, %9, " r
#define P_AB_ACC(r) ", %8, %9, " r
#define P_A_ACC(r) ", %8, " r
#define P_ACC_ONLY1(r) ", 1, " r
#define P_LSHLADD(r) ", " r ", 0, " r
#define P_FMA(r) ", " r ", " r ", " r
#define P_ALIGN(r) ", %8, " r ", 28"
#define P_ANDOR(r) ", " r ", %8, %9"
#define P_BFE(r) ", " r ", 4, 28"
#define P_ADDC(r) ", vcc, %8, " r ", vcc"
#define P_ADDCO(r) ", vcc, %8, " r
#define P_MOV(r) ", %8"

ALL8(k_mad_u64_u32, uint64_t, "v_mad_u64_u32 ", P_MAD)
ALL8(k_mad_i64_i32, uint64_t, "v_mad_i64_i32 ", P_MAD)
ALL8(k_lshl_add_u64, uint64_t, "v_lshl_add_u64 ", P_LSHLADD)
ALL8(k_lshrrev_b64, uint64_t, "v_lshrrev_b64 ", P_ACC_ONLY1)
ALL8(k_fma_f64, uint64_t, "v_fma_f64 ", P_FMA)
ALL8(k_pk_fma_f32, uint64_t, "v_pk_fma_f32 ", P_FMA)
ALL8(k_mul_lo_u32, uint32_t, "v_mul_lo_u32 ", P_A_ACC)
ALL8(k_mul_hi_u32, uint32_t, "v_mul_hi_u32 ", P_A_ACC)
ALL8(k_mad_u32_u24, uint32_t, "v_mad_u32_u24 ", P_AB_ACC)
ALL8(k_mul_u32_u24, uint32_t, "v_mul_u32_u24_e32 ", P_A_ACC)
ALL8(k_add_u32, uint32_t, "v_add_u32_e32 ", P_A_ACC)
ALL8(k_sub_u32, uint32_t, "v_sub_u32_e32 ", P_A_ACC)
ALL8(k_add3_u32, uint32_t, "v_add3_u32 ", P_AB_ACC)
ALL8(k_and_b32, uint32_t, "v_and_b32_e32 ", P_A_ACC)
ALL8(k_and_or_b32, uint32_t, "v_and_or_b32 ", P_ANDOR)
ALL8(k_lshrrev_b32, uint32_t, "v_lshrrev_b32_e32 ", P_ACC_ONLY1)
ALL8(k_alignbit_b32, uint32_t, "v_alignbit_b32 ", P_ALIGN)
ALL8(k_bfe_u32, uint32_t, "v_bfe_u32 ", P_BFE)
ALL8(k_add_co_u32, uint32_t, "v_add_co_u32_e32 ", P_ADDCO)
ALL8(k_addc_co_u32, uint32_t, "v_addc_co_u32_e32 ", P_ADDC)
ALL8(k_fma_f32, uint32_t, "v_fma_f32 ", P_FMA)
ALL8(k_mov_b32, uint32_t, "v_mov_b32_e32 ", P_MOV)
// the same simple ops carrying a 32-bit literal (8-byte encoding), as the limb masks and subtraction
// biases of the field arithmetic do
#define P_LIT_MASK(r) ", 0xfffffff, " r
#define P_LIT_BIAS(r) ", 0x3ffffffc, " r
ALL8(k_and_b32_literal, uint32_t, "v_and_b32_e32 ", P_LIT_MASK)
ALL8(k_add_u32_literal, uint32_t, "v_add_u32_e32 ", P_LIT_BIAS)

// instruction mixes of the Ed448 ladder: MAC:simple = 1:1 and the doubling's measured histogram
DEF_KERNEL(k_mix_mac_add, uint64_t, "v_mad_u64_u32 %0, vcc, %8, %9, %0", "v_add_u32_e32 %8, %9, %8",
           "v_mad_u64_u32 %1, vcc, %8, %9, %1", "v_and_b32_e32 %9, %8, %9", "v_mad_u64_u32 %2, vcc, %8, %9, %2",
           "v_add_u32_e32 %8, %9, %8", "v_mad_u64_u32 %3, vcc, %8, %9, %3", "v_and_b32_e32 %9, %8, %9")
DEF_KERNEL(k_mix_mac3_add, uint64_t, "v_mad_u64_u32 %0, vcc, %8, %9, %0", "v_mad_u64_u32 %1, vcc, %8, %9, %1",
           "v_mad_u64_u32 %2, vcc, %8, %9, %2", "v_add_u32_e32 %8, %9, %8", "v_mad_u64_u32 %3, vcc, %8, %9, %3",
           "v_mad_u64_u32 %4, vcc, %8, %9, %4", "v_mad_u64_u32 %5, vcc, %8, %9, %5", "v_and_b32_e32 %9, %8, %9")
DEF_KERNEL(k_mix_mac_lshladd, uint64_t, "v_mad_u64_u32 %0, vcc, %8, %9, %0", "v_lshl_add_u64 %4, %4, 0, %5",
           "v_mad_u64_u32 %1, vcc, %8, %9, %1", "v_lshl_add_u64 %5, %5, 0, %6", "v_mad_u64_u32 %2, vcc, %8, %9, %2",
           "v_lshl_add_u64 %6, %6, 0, %7", "v_mad_u64_u32 %3, vcc, %8, %9, %3", "v_lshl_add_u64 %7, %7, 0, %4")

// Does the order matter?  Same 16 MACs + 16 simple ops per block on independent registers,
// (a) strictly alternating, (b) clustered in runs of 8, (c) clustered in runs of 16.
#define MIXK(NAME, BODY)                                                                           \
    template <int K>                                                                               \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t a0, uint32_t b0,           \
                                                unsigned long long *cyc) {                         \
        uint64_t m0, m1, m2, m3;                                                                   \
        uint32_t s0, s1, s2, s3;                                                                   \
        uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
        m0 = a; m1 = a * 2; m2 = a * 3; m3 = a * 4; s0 = b; s1 = b * 3; s2 = b * 5; s3 = b * 7;     \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITERS; it++) {                                                       \
            asm volatile(BODY : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) \
                         : "v"(a), "v"(b) : "vcc");                                                \
        }                                                                                          \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = m0 ^ m1 ^ m2 ^ m3 ^ s0 ^ s1 ^ s2 ^ s3;        \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
    }
#define M0 "v_mad_u64_u32 %0, vcc, %8, %9, %0\n"
#define M1 "v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
#define M2 "v_mad_u64_u32 %2, vcc, %8, %9, %2\n"
#define M3 "v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
#define S0 "v_add_u32_e32 %4, %8, %4\n"
#define S1 "v_and_b32_e32 %5, %9, %5\n"
#define S2 "v_add_u32_e32 %6, %8, %6\n"
#define S3 "v_lshrrev_b32_e32 %7, 1, %7\n"
// a 1:1 mix whose simple ops carry literals / whose simple ops read the MAC result just written
#define L0 "v_add_u32_e32 %4, 0x3ffffffc, %4\n"
#define L1 "v_and_b32_e32 %5, 0xfffffff, %5\n"
#define L2 "v_add_u32_e32 %6, 0x3ffffffc, %6\n"
#define L3 "v_and_b32_e32 %7, 0xfffffff, %7\n"
MIXK(k_mix_runs4_literal, ".rept 4\n" M0 M1 M2 M3 L0 L1 L2 L3 "\n.endr")
MIXK(k_ord_alternating, ".rept 4\n" M0 S0 M1 S1 M2 S2 M3 S3 "\n.endr")
MIXK(k_ord_runs_of_4, ".rept 4\n" M0 M1 M2 M3 S0 S1 S2 S3 "\n.endr")
MIXK(k_ord_runs_of_16, ".rept 4\n" M0 M1 M2 M3 "\n.endr\n.rept 4\n" S0 S1 S2 S3 "\n.endr")
MIXK(k_ord_runs_of_64, ".rept 16\n" M0 M1 M2 M3 "\n.endr\n.rept 16\n" S0 S1 S2 S3 "\n.endr")
MIXK(k_ord_runs_of_256, ".rept 64\n" M0 M1 M2 M3 "\n.endr\n.rept 64\n" S0 S1 S2 S3 "\n.endr")
MIXK(k_ord_mac_only16, ".rept 4\n" M0 M1 M2 M3 "\n.endr")
MIXK(k_ord_simple_only16, ".rept 4\n" S0 S1 S2 S3 "\n.endr")

typedef void (*kern_t)(uint64_t *, uint32_t, uint32_t, unsigned long long *);

struct Entry {
    const char *name;
    kern_t k1, k8;
    int insts;   // instructions per loop iteration
};
#define ENT(n) {#n, n<1>, n<8>, UNROLL}
#define ENTN(n, k) {#n, n<1>, n<8>, k}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    Entry ents[] = {ENT(k_mad_u64_u32), ENT(k_mad_i64_i32), ENT(k_mul_lo_u32), ENT(k_mul_hi_u32),
                    ENT(k_mad_u32_u24), ENT(k_mul_u32_u24), ENT(k_add_u32), ENT(k_sub_u32), ENT(k_add3_u32),
                    ENT(k_and_b32), ENT(k_and_or_b32), ENT(k_bfe_u32), ENT(k_lshrrev_b32), ENT(k_alignbit_b32),
                    ENT(k_add_co_u32), ENT(k_addc_co_u32), ENT(k_mov_b32), ENT(k_lshl_add_u64), ENT(k_lshrrev_b64),
                    ENT(k_fma_f32), ENT(k_pk_fma_f32), ENT(k_fma_f64), ENT(k_mix_mac_add), ENT(k_mix_mac3_add),
                    ENT(k_mix_mac_lshladd), ENT(k_ord_alternating), ENT(k_ord_runs_of_4),
                    ENT(k_ord_runs_of_16), ENTN(k_ord_mac_only16, 16), ENTN(k_ord_simple_only16, 16),
                    ENTN(k_ord_runs_of_64, 128), ENTN(k_ord_runs_of_256, 512), ENT(k_and_b32_literal), ENT(k_add_u32_literal), ENT(k_mix_runs4_literal)};
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
                double insts_per_wave = (double)ITERS * e.insts;
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
