// fp64gate.hip -- the gate of VERDICT r04 item 1: can a field multiplication on FP64 FMAs (v_fma_f64) beat the
// 28-bit-limb / v_mad_u64_u32 one on gfx950?  This tool times LOWER BOUNDS of the two FP64 formulations there are --
// the multiply instructions of one field multiplication and nothing else: no conversions, no carries, no reduction --
// against the COMPLETE integer multiplication (gf28.hpp fe_mul, fe_sqr; gf28s.hpp's signed ones), on the whole chip at
// two waves per SIMD, the way tools/fieldbench does.  docs/fp64_gate.md has the derivation and the verdict.
//
//   (A) exact products by two FMAs (Emmart-Weems): hi = fma(a, b, C1); lo = fma(a, b, C2 - hi); the two halves are
//       accumulated as INTEGERS (their mantissas), so a limb product costs fma + sub + fma + two 64-bit additions
//       = 5 instructions for 50 x 50 bits.  9 limbs of 50 bits, schoolbook: 81 products; 10 limbs of 45 bits with
//       Karatsuba over phi: 75.
//   (B) hybrid: the HIGH part of a column by one chained FMA per product (an accumulator anchored at 2^(52+s) keeps
//       ulp 2^s, so every step adds the product truncated to a multiple of 2^s, exactly), the LOW part from the column
//       sum mod 2^32, one chained v_mad_u64_u32 on the low words per product: 2 instructions per product, but
//       s + log2(terms) <= 32 and 2w + log2(terms) <= 52 + s bound the limbs to w <= 38 bits: 12 limbs, 108 products
//       with Karatsuba over phi = 216 multiply instructions (the integer multiplication has 192).
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -o tools/fp64gate tools/fp64gate.hip
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include "gf28s.hpp"

using namespace gd;

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

enum Op { INT_MUL, INT_SQR, SGN_MUL, SGN_SQR, FP_A_81, FP_A_75, FP_B_108, FP_B_108_READOUT };

// (A): PRODUCTS exact limb products, three independent accumulation chains at a time (like the integer code's)
template <int PRODUCTS>
__device__ __forceinline__ void fp_a(double (&a)[10], double (&b)[10], uint64_t (&sum_hi)[3], uint64_t (&sum_lo)[3]) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
#pragma unroll
    for (int k = 0; k < PRODUCTS; k++) {
        const double x = a[k % 10], y = b[(k / 10 + k) % 10];   // 10 x 10 distinct pairs
        const double hi = __builtin_fma(x, y, C1);
        const double sub = C2 - hi;
        const double lo = __builtin_fma(x, y, sub);
        sum_hi[k % 3] += (uint64_t)__double_as_longlong(hi);
        sum_lo[k % 3] += (uint64_t)__double_as_longlong(lo);
    }
}

// (B): PRODUCTS times { one FMA onto an anchored accumulator, one v_mad_u64_u32 on the low words }, three chains
template <int PRODUCTS>
__device__ __forceinline__ void fp_b(double (&a)[12], double (&b)[12], uint32_t (&al)[12], uint32_t (&bl)[12], double (&acc)[3],
                                     uint64_t (&low)[3]) {
#pragma unroll
    for (int k = 0; k < PRODUCTS; k++) {
        acc[k % 3] = __builtin_fma(a[k % 12], b[(k / 12 + k) % 12], acc[k % 3]);   // 12 x 12 distinct pairs
        low[k % 3] += (uint64_t)al[k % 12] * bl[(k / 12 + k) % 12];
        asm("" : "+v"(acc[k % 3]), "+v"(low[k % 3]));
    }
}

// (B) again, plus what every one of its 24 columns and 12 result limbs needs at the very least before the next
// multiplication can start -- still no Karatsuba combination, no mixed-radix doubling, no reduction by p:
//   column:  H = acc - anchor (v_add_f64); its integer bits by a magic addition (v_add_f64); E = low word - H's low word
//            (v_sub_u32); the column = H + E (v_lshl_add_u64 with E sign-extended: + v_ashrrev_i32); + carry in
//            (v_lshl_add_u64); limb = low 38 bits (two v_and); carry out (v_lshrrev_b64)                      = 9
//   result limb: back to a double (two v_cvt_f64_u32 + one v_fma_f64) and its low word (free)                  = 3
__device__ __forceinline__ void fp_b_readout(double (&a)[12], uint32_t (&al)[12], double (&acc)[3], uint64_t (&low)[3]) {
    uint64_t carry = 0;
#pragma unroll
    for (int c = 0; c < 24; c++) {
        double h = acc[c % 3] - 0x1p84;
        double m = h + 0x1.8p52;
        asm("" : "+v"(h), "+v"(m));
        const uint64_t hb = (uint64_t)__double_as_longlong(m) & 0xfffffffffffffull;
        const int32_t e = (int32_t)((uint32_t)low[c % 3] - (uint32_t)hb);
        uint64_t col = hb + (uint64_t)(int64_t)e + carry;
        asm("" : "+v"(col));
        const uint64_t limb = col & ((1ull << 38) - 1);
        carry = col >> 38;
        if (c < 12) {
            const uint32_t lo = (uint32_t)limb, hi = (uint32_t)(limb >> 32);
            a[c] = __builtin_fma((double)hi, 0x1p32, (double)lo);
            al[c] = lo;
        } else {
            low[c % 3] += limb;   // (keeps the high columns' read-outs alive)
        }
    }
    acc[0] += (double)(uint32_t)carry;
}

template <int OP>
__global__ void __launch_bounds__(256, 2) k_chain(uint32_t *io, int n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x, y;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x.v[i] = io[t * 32 + i] & M28;
        y.v[i] = io[t * 32 + 16 + i] & M28;
    }
    if (OP == INT_MUL || OP == INT_SQR) {
        for (int k = 0; k < n; k++) x = OP == INT_MUL ? fe_mul(x, y) : fe_sqr(x);
    } else if (OP == SGN_MUL || OP == SGN_SQR) {
        sfp sx = sfe_from_fe(x);
        const sfp sy = sfe_from_fe(y);
        for (int k = 0; k < n; k++) sx = OP == SGN_MUL ? sfe_mul(sx, sy) : sfe_sqr<false>(sx);
        x = sfe_to_fe(sx);
    } else if (OP == FP_A_81 || OP == FP_A_75) {
        double a[10], b[10];
        uint64_t sh[3] = {0, 0, 0}, sl[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 10; i++) {
            a[i] = (double)x.v[i] * 4194304.0;   // ~50-bit integers
            b[i] = (double)y.v[i] * 4194304.0;
        }
        for (int k = 0; k < n; k++) {
#pragma unroll
            for (int i = 0; i < 10; i++) asm("" : "+v"(a[i]), "+v"(b[i]));   // new operands every round, as far as the compiler knows
            if (OP == FP_A_81) fp_a<81>(a, b, sh, sl);
            else fp_a<75>(a, b, sh, sl);
        }
        x.v[0] = (uint32_t)(sh[0] + sh[1] + sh[2]);
        x.v[1] = (uint32_t)(sl[0] + sl[1] + sl[2]);
    } else {   // FP_B_108, FP_B_108_READOUT
        double a[12], b[12], acc[3] = {0x1p84, 0x1p84, 0x1p84};
        uint32_t al[12], bl[12];
        uint64_t low[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 12; i++) {
            a[i] = (double)x.v[i] * 1024.0;      // 38-bit integers
            b[i] = (double)y.v[i] * 1024.0;
            al[i] = x.v[i] << 10;
            bl[i] = y.v[i] << 10;
        }
        for (int k = 0; k < n; k++) {
#pragma unroll
            for (int i = 0; i < 12; i++) asm("" : "+v"(a[i]), "+v"(b[i]), "+v"(al[i]), "+v"(bl[i]));
            fp_b<108>(a, b, al, bl, acc, low);
            if (OP == FP_B_108_READOUT) fp_b_readout(a, al, acc, low);
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] = acc[c] > 0x1p85 ? 0x1p84 : acc[c];   // (stay in the anchored binade)
        }
        x.v[0] = (uint32_t)(low[0] + low[1] + low[2]);
        x.v[1] = (uint32_t)__double_as_longlong(acc[0] + acc[1] + acc[2]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) io[t * 32 + i] = x.v[i];
}

template <int OP>
static void run(const char *name, uint32_t *d_io) {
    const int n = 4000, w = 2;
    const int blocks = 256 * w;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<OP>, dim3(blocks), dim3(256), 0, 0, d_io, 16);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain<OP>, dim3(blocks), dim3(256), 0, 0, d_io, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-58s %8.3f ms  %8.1f SIMD-cycles per wave-op at 2 waves/SIMD (nominal 2.4 GHz)\n", name, best,
           best * 1e-3 * 2.4e9 / n / w);
}

int main() {
    const size_t lanes = 256 * 2 * 256;
    uint32_t *h = (uint32_t *)malloc(lanes * 32 * 4), *d;
    for (size_t i = 0; i < lanes * 32; i++) h[i] = (uint32_t)(i * 2654435761u) >> 3;
    CHECK(hipMalloc(&d, lanes * 32 * 4));
    CHECK(hipMemcpy(d, h, lanes * 32 * 4, hipMemcpyHostToDevice));
    for (int pass = 0; pass < 2; pass++) {
        run<INT_MUL>("integer fe_mul, complete (192 MACs + 82)", d);
        run<INT_SQR>("integer fe_sqr, complete (136 MACs + 78)", d);
        run<SGN_MUL>("signed sfe_mul, complete (192 MACs)", d);
        run<SGN_SQR>("signed sfe_sqr, complete (136 MACs)", d);
        run<FP_A_81>("FP64 (A) 81 exact products, multiply instructions ONLY", d);
        run<FP_A_75>("FP64 (A) 75 exact products, multiply instructions ONLY", d);
        run<FP_B_108>("FP64+int (B) 108 products, multiply instructions ONLY", d);
        run<FP_B_108_READOUT>("FP64+int (B) 108 products + column read-out + conversion", d);
    }
    return 0;
}
