// fieldbench.hip -- A/B timing of the lane field operations on the whole chip (gfx950).
//
// Each lane runs a dependent chain of N field operations (x = op(x, y)); the grid puts W waves on
// every SIMD (256 CUs x 4 SIMDs).  Reported: SIMD cycles per wave-operation at the nominal clock
// (wall time x 2.4 GHz x W / N) -- the quantity the ladders are bound by -- next to the static
// MAC / other-VALU instruction counts, so that candidate formulations of gf_mul / gf_sqr can be
// compared on the same box in the same run.  Variants that lost are kept here (not in the
// library) as the evidence for the choice made in gf28.hpp.
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -o tools/fieldbench tools/fieldbench.hip
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include "gf28.hpp"
#include "point.hpp"

using namespace gd;

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

// ---- variant: the same square as the library's with the cross terms doubled on the OPERANDS (2x arrays, 24
// shifts per square) instead of once per finished column -- the library's version until round 2
namespace opdbl {
struct sq8 {
    uint32_t x[8], x2[8];
};
template <int COL>
GD_FN void sq_col(acc_t &acc, const sq8 &s) {  // acc += column COL (0..14) of x^2
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7 || j > k) continue;
        if (j == k) acc.mac(s.x[j], s.x[j]);
        else acc.mac(s.x2[j], s.x[k]);
    }
}
template <int I>
GD_FN void sqr_column(fe &c, acc_t &lo, acc_t &hi, const sq8 &u, const sq8 &v, const sq8 &s, const uint32_t (&t)[8]) {
    sq_col<I>(lo, u);
    sq_col<I>(lo, v);
    mul_col<I>(hi, v.x, t);
    if (I < 7) {
        mul_col<I + 8>(lo, v.x, t);
        sq_col<I + 8>(hi, s);
        sq_col<I + 8>(hi, v);
    }
    c.v[I] = lo.lo28();
    c.v[I + 8] = hi.lo28();
    lo.shr28();
    hi.shr28();
}
GD_FN fe fe_sqr(const fe &a) {
    sq8 u, v, s;
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u.x[j] = a.v[j];
        v.x[j] = a.v[j + 8];
        s.x[j] = a.v[j] + a.v[j + 8];
        u.x2[j] = u.x[j] << 1;
        v.x2[j] = v.x[j] << 1;
        s.x2[j] = s.x[j] << 1;
        t[j] = u.x2[j] + v.x[j];
    }
    fe c;
    acc_t lo, hi;
    sqr_column<0>(c, lo, hi, u, v, s, t);
    sqr_column<1>(c, lo, hi, u, v, s, t);
    sqr_column<2>(c, lo, hi, u, v, s, t);
    sqr_column<3>(c, lo, hi, u, v, s, t);
    sqr_column<4>(c, lo, hi, u, v, s, t);
    sqr_column<5>(c, lo, hi, u, v, s, t);
    sqr_column<6>(c, lo, hi, u, v, s, t);
    sqr_column<7>(c, lo, hi, u, v, s, t);
    fe_fold_tails(c, lo, hi);
    return c;
}
}  // namespace opdbl

// ---- variant: Karatsuba square, 108 MACs + 7 64-bit add/sub per column pair (round-1 original)
namespace kar {
using opdbl::sq8;
using opdbl::sq_col;
template <int I>
GD_FN void sqr_column(fe &c, acc_t &lo, acc_t &hi, const sq8 &u, const sq8 &v, const sq8 &s) {
    acc_t A, Cw, E;
    sq_col<I>(A, u);
    if (I < 7) {
        sq_col<I + 8>(Cw, u);
        sq_col<I + 8>(E, s);
        sq_col<I + 8>(hi, v);
    }
    sq_col<I>(lo, v);
    sq_col<I>(hi, s);
    lo.add(A);
    if (I < 7) {
        lo.add(E);
        lo.sub(Cw);
        hi.add(E);
    }
    hi.sub(A);
    c.v[I] = lo.lo28();
    c.v[I + 8] = hi.lo28();
    lo.shr28();
    hi.shr28();
}
GD_FN fe fe_sqr(const fe &a) {
    sq8 u, v, s;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u.x[j] = a.v[j];
        v.x[j] = a.v[j + 8];
        s.x[j] = a.v[j] + a.v[j + 8];
        u.x2[j] = u.x[j] << 1;
        v.x2[j] = v.x[j] << 1;
        s.x2[j] = s.x[j] << 1;
    }
    fe c;
    acc_t lo, hi;
    sqr_column<0>(c, lo, hi, u, v, s);
    sqr_column<1>(c, lo, hi, u, v, s);
    sqr_column<2>(c, lo, hi, u, v, s);
    sqr_column<3>(c, lo, hi, u, v, s);
    sqr_column<4>(c, lo, hi, u, v, s);
    sqr_column<5>(c, lo, hi, u, v, s);
    sqr_column<6>(c, lo, hi, u, v, s);
    sqr_column<7>(c, lo, hi, u, v, s);
    fe_fold_tails(c, lo, hi);
    return c;
}
}  // namespace kar

// ---- variant: schoolbook-over-phi multiplication, 256 MACs, no 64-bit combine arithmetic:
//   low_i  = (a0 b0 + a1 b1)_i + (a0 b1 + a1 sb)_i'      high_i = (a0 b1 + a1 sb)_i + (sa sb + a1 b1)_i'
namespace direct {
GD_FN fe fe_mul(const fe &a, const fe &b) {
    uint32_t a0[8], a1[8], b0[8], b1[8], sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        a0[j] = a.v[j]; a1[j] = a.v[j + 8]; b0[j] = b.v[j]; b1[j] = b.v[j + 8];
        sa[j] = a0[j] + a1[j];
        sb[j] = b0[j] + b1[j];
    }
    fe c;
    acc_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j <= i) {
                lo.mac(a0[j], b0[i - j]);
                lo.mac(a1[j], b1[i - j]);
                hi.mac(a0[j], b1[i - j]);
                hi.mac(a1[j], sb[i - j]);
            } else {
                lo.mac(a0[j], b1[i - j + 8]);
                lo.mac(a1[j], sb[i - j + 8]);
                hi.mac(sa[j], sb[i - j + 8]);
                hi.mac(a1[j], b1[i - j + 8]);
            }
        }
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    fe_fold_tails(c, lo, hi);
    return c;
}
}  // namespace direct

// ---- prototype (timing only): SIGNED limbs.  Differences need no bias (a - b instead of a + K p - b: one
// instruction per limb instead of two), products are v_mad_i64_i32, carries arithmetic shifts.  The doubling
// below has the library's shape (4 squarings, 3 products, one weak reduction) without the three bias additions.
namespace sgn {
struct sfe {
    int32_t v[16];
};
struct sacc {
    int64_t x = 0;
    GD_MFN void mac(int32_t a, int32_t b) {
        x += (int64_t)a * b;
        asm("" : "+v"(x));
    }
    GD_MFN void add(const sacc &o) { x += o.x; }
    GD_MFN void add_doubled(const sacc &o) { x += o.x << 1; }
    GD_MFN void sub(const sacc &o) { x -= o.x; }
    GD_MFN int32_t lo28() const { return (int32_t)((uint32_t)x & M28); }
    GD_MFN void shr28() { x >>= 28; }
};
GD_FN void fold_tails(sfe &c, sacc lo, sacc hi) {
    lo.add(hi);
    lo.x += c.v[8];
    hi.x += c.v[0];
    c.v[8] = lo.lo28();
    c.v[0] = hi.lo28();
    lo.shr28();
    hi.shr28();
    c.v[9] += (int32_t)lo.x;
    c.v[1] += (int32_t)hi.x;
}
GD_FN sfe mul(const sfe &a, const sfe &b) {
    int32_t sa[8], sb[8], sbb[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sa[j] = a.v[j] + a.v[j + 8];
        sb[j] = b.v[j] + b.v[j + 8];
        sbb[j] = sb[j] + b.v[j + 8];
    }
    sfe c;
    sacc lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        sacc cross;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j <= i) {
                cross.mac(a.v[j], b.v[i - j]);
                hi.mac(sa[j], sb[i - j]);
                lo.mac(a.v[j + 8], b.v[i - j + 8]);
            } else {
                cross.mac(a.v[j], b.v[i - j + 16]);
                hi.mac(sa[j], sbb[i - j + 8]);
                lo.mac(a.v[j + 8], sb[i - j + 8]);
            }
        }
        hi.sub(cross);
        lo.add(cross);
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    fold_tails(c, lo, hi);
    return c;
}
template <int COL>
GD_FN void sq_col(sacc &cross, sacc &rest, const int32_t (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7 || j > k) continue;
        if (j == k) rest.mac(x[j], x[j]);
        else cross.mac(x[j], x[k]);
    }
}
template <int COL>
GD_FN void mul_col(sacc &acc, const int32_t (&x)[8], const int32_t (&y)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7) continue;
        acc.mac(x[j], y[k]);
    }
}
template <int I>
GD_FN void sqr_column(sfe &c, sacc &lo, sacc &hi, const int32_t (&u)[8], const int32_t (&v)[8], const int32_t (&s)[8],
                      const int32_t (&t)[8]) {
    sacc lo_cross, hi_cross;
    sq_col<I>(lo_cross, lo, u);
    sq_col<I>(lo_cross, lo, v);
    mul_col<I>(hi, v, t);
    if (I < 7) {
        mul_col<I + 8>(lo, v, t);
        sq_col<I + 8>(hi_cross, hi, s);
        sq_col<I + 8>(hi_cross, hi, v);
    }
    if (I > 0) lo.add_doubled(lo_cross);
    if (I < 7) hi.add_doubled(hi_cross);
    c.v[I] = lo.lo28();
    c.v[I + 8] = hi.lo28();
    lo.shr28();
    hi.shr28();
}
GD_FN sfe sqr(const sfe &a) {
    int32_t u[8], v[8], s[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u[j] = a.v[j];
        v[j] = a.v[j + 8];
        s[j] = a.v[j] + a.v[j + 8];
        t[j] = (a.v[j] << 1) + a.v[j + 8];
    }
    sfe c;
    sacc lo, hi;
    sqr_column<0>(c, lo, hi, u, v, s, t);
    sqr_column<1>(c, lo, hi, u, v, s, t);
    sqr_column<2>(c, lo, hi, u, v, s, t);
    sqr_column<3>(c, lo, hi, u, v, s, t);
    sqr_column<4>(c, lo, hi, u, v, s, t);
    sqr_column<5>(c, lo, hi, u, v, s, t);
    sqr_column<6>(c, lo, hi, u, v, s, t);
    sqr_column<7>(c, lo, hi, u, v, s, t);
    fold_tails(c, lo, hi);
    return c;
}
GD_FN sfe add(const sfe &a, const sfe &b) {
    sfe c;
#pragma unroll
    for (int i = 0; i < 16; i++) c.v[i] = a.v[i] + b.v[i];
    return c;
}
GD_FN sfe sub(const sfe &a, const sfe &b) {
    sfe c;
#pragma unroll
    for (int i = 0; i < 16; i++) c.v[i] = a.v[i] - b.v[i];
    return c;
}
GD_FN sfe weak(const sfe &a) {
    sfe c;
    const int32_t top = a.v[15] >> 28;
    c.v[0] = (a.v[0] & (int32_t)M28) + top;
#pragma unroll
    for (int i = 1; i < 16; i++) c.v[i] = (a.v[i] & (int32_t)M28) + (a.v[i - 1] >> 28);
    c.v[8] += top;
    return c;
}
struct spt {
    sfe x, y, z, t;
};
GD_FN void dbl(spt &p) {
    sfe c = sqr(p.x);
    sfe a = sqr(p.y);
    sfe d = add(c, a);
    sfe s = add(p.x, p.y);
    sfe b = sub(sqr(s), d);
    sfe tt = sub(a, c);
    sfe zz = sqr(p.z);
    sfe e = weak(sub(add(zz, zz), tt));
    p.x = mul(e, b);
    p.z = mul(e, tt);
    p.y = mul(d, tt);
}
}  // namespace sgn

enum Op { MUL, SQR, SQR_KAR, MUL_DIRECT, DBL, ADD_WEAK, SQR_OPDBL, DBL_SIGNED, ISR };

template <int OP>
__global__ void __launch_bounds__(256, 2) k_chain(uint32_t *io, int n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x, y;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x.v[i] = io[t * 32 + i] & M28;
        y.v[i] = io[t * 32 + 16 + i] & M28;
    }
    if (OP == DBL_SIGNED) {
        sgn::spt p;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            p.x.v[i] = (int32_t)x.v[i];
            p.y.v[i] = (int32_t)y.v[i];
            p.z.v[i] = (int32_t)((x.v[i] + y.v[i]) & M28);
        }
        for (int k = 0; k < n; k++) sgn::dbl(p);
#pragma unroll
        for (int i = 0; i < 16; i++) x.v[i] = (uint32_t)(p.x.v[i] + p.y.v[i] + p.z.v[i]);
    } else if (OP == DBL) {
        pt p;
        p.x = x; p.y = y; p.z = fe_add(x, y); p.z = fe_weak(p.z); p.t = x;
        for (int k = 0; k < n; k++) pt_double(p, false);
        x = fe_weak(fe_add(fe_add(p.x, p.y), p.z));
    } else {
        for (int k = 0; k < n; k++) {
            if (OP == MUL) x = fe_mul(x, y);
            if (OP == SQR) x = fe_sqr(x);
            if (OP == SQR_KAR) x = kar::fe_sqr(x);
            if (OP == SQR_OPDBL) x = opdbl::fe_sqr(x);
            if (OP == MUL_DIRECT) x = direct::fe_mul(x, y);
            if (OP == ADD_WEAK) x = fe_weak(fe_add(x, y));
            if (OP == ISR) {   // the whole exponentiation of a decoding: 446 squarings + 13 multiplications, in its loops
                bool ok;
                x = fe_isr(x, &ok);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) io[t * 32 + i] = x.v[i];
}

template <int OP>
static void run(const char *name, uint32_t *d_io, int macs, int waves_per_simd) {
    const int n = OP == ISR ? 10 : 4000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs, 4 waves per block = 1 per SIMD
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
    // every SIMD runs waves_per_simd chains of n ops concurrently
    double cyc = best * 1e-3 * 2.4e9 / n / waves_per_simd;
    printf("%-12s w/SIMD=%d  %8.3f ms  %8.1f SIMD-cycles per wave-op", name, waves_per_simd, best, cyc);
    if (macs) printf("  (%d MACs -> %.2f cycles per MAC-equivalent)", macs, cyc / macs);
    printf("\n");
}

int main() {
    const size_t lanes = 256 * 2 * 256;
    uint32_t *h = (uint32_t *)malloc(lanes * 32 * 4), *d;
    for (size_t i = 0; i < lanes * 32; i++) h[i] = (uint32_t)(i * 2654435761u) >> 3;
    CHECK(hipMalloc(&d, lanes * 32 * 4));
    CHECK(hipMemcpy(d, h, lanes * 32 * 4, hipMemcpyHostToDevice));
    for (int w = 1; w <= 2; w++) {
        run<MUL>("mul", d, 192, w);
        run<MUL_DIRECT>("mul_direct", d, 256, w);
        run<SQR>("sqr", d, 136, w);
        run<SQR_KAR>("sqr_kar", d, 108, w);
        run<SQR_OPDBL>("sqr_opdbl", d, 136, w);
        run<DBL>("pt_double", d, 4 * 136 + 3 * 192, w);
        run<DBL_SIGNED>("dbl_signed", d, 4 * 136 + 3 * 192, w);
        run<ADD_WEAK>("add+weak", d, 0, w);
        run<ISR>("isr", d, 63152, w);
    }
    return 0;
}
