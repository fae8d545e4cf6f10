// gf28.hpp -- GF(2^448 - 2^224 - 1) for one element per wavefront lane on gfx950.
//
// Device representation: 16 limbs of 28 bits held in 16 VGPRs (radix 2^28).  The
// reference's ABI representation (8 x 56-bit limbs, src/f_field.h:23-29) is
// converted at kernel load/store only.  28-bit limbs map 1:1 onto CDNA4's
// v_mad_u64_u32 (32x32+64 -> 64) and leave 4 bits of per-limb headroom so that
// sums/differences of a few elements need no carry propagation ("lazy" adds).
//
// The multiplication restates the identity behind the reference's gf_mul
// (src/arch_ref64/f_impl.c:7-166; math in SURVEY.md section 9): with phi = 2^224,
// a = a0 + a1*phi, b = b0 + b1*phi and phi^2 = phi + 1,
//     a*b = (a0*b0 + a1*b1) + ((a0+a1)*(b0+b1) - a0*b0) * phi     (mod p)
// so three 8x8-limb half products (192 MACs) replace the 256 of schoolbook.
//
// MAGNITUDE CONTRACT (checked in the host-side checker build, GF_CHECKED):
//   "mag k" = every limb <= k * 2^28 (+ a few units).  mul/sqr results are mag 1
//   (limbs < 2^28, limbs 1 and 9 < 2^28 + 2^10).  The accumulators are u64 and wrap: only the
//   FINISHED column values (after Karatsuba's subtraction) have to fit 64 bits, intermediate
//   chain values may exceed 2^64 because every step is exact mod 2^64.  The largest finished
//   column of a product is high_0 = 3 + 7*5 = 38 limb products, so mul(a,b) needs
//   38 * maxlimb(a) * maxlimb(b) + 2^37 < 2^64, i.e. mag(a)*mag(b) <= 6.7 (sum x difference = 2 x 3
//   is fine, difference x difference = 9 is not), and mag(a) <= 7, mag(b) <= 5 so the pre-added
//   halves (a0+a1, b0+2*b1) fit 32 bits.  sqr(a): 38 * maxlimb(a)^2, mag(a) <= 2.5.
//
// This header compiles for the device with hipcc and, for tests/hostsim only,
// as plain C++ with g++ (the product never runs the host build).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GD_FN __device__ __forceinline__
#define GD_MFN __device__ __forceinline__
// on a lambda handed to a loop helper: hipcc otherwise leaves a large body out of line -- a call per operation, the
// captured state through scratch (k_ed448_verify_keycomb: 4 % of the kernel)
#define GD_LAMBDA_INLINE __attribute__((always_inline))
#define GD_CONST __device__ const
#else
#define GD_FN static inline   // the host checker build lets g++ decide (forced inlining costs minutes of compile time)
#define GD_MFN inline
#define GD_LAMBDA_INLINE
#define GD_CONST static const
#endif

namespace gd {

// Nothing is scheduled across this point (device build; nothing on the host): used where a memory read has to be
// ISSUED early -- ahead of a block of arithmetic it does not depend on -- and the compiler's scheduler, minding
// register pressure, would move it down to its first use.
GD_FN void gd_keep_order() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

constexpr uint32_t M28 = (1u << 28) - 1;

struct fe {
    uint32_t v[16];
};

// (a << SH) + b on the limb pair (lo, hi) = limbs 2k, 2k+1 with ONE v_lshl_add_u64: the two limbs are the halves of a
// 64-bit register pair, and as the magnitude contract keeps every limb (and every sum of limbs formed here) below 2^32,
// the low half never carries into the high half.  Every VALU instruction of these kernels costs the same four cycles
// (DESIGN.md section 7), so an addition of two elements is 8 instructions instead of 16.  The instruction is written out
// and its operands are opaque: left to itself the compiler takes the 64-bit addition apart again.
template <int SH>
GD_FN void fe_pair_add(uint32_t &lo, uint32_t &hi, uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi) {
#if defined(GF_CHECKED)
    if ((((uint64_t)alo << SH) + blo) >> 32 || (((uint64_t)ahi << SH) + bhi) >> 32) __builtin_trap();
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GD_NO_PAIRED_ADDS)
    const uint64_t a = (uint64_t)alo | ((uint64_t)ahi << 32), b = (uint64_t)blo | ((uint64_t)bhi << 32);
    uint64_t c;
    asm("v_lshl_add_u64 %0, %1, %3, %2" : "=v"(c) : "v"(a), "v"(b), "n"(SH));
    lo = (uint32_t)c;
    hi = (uint32_t)(c >> 32);
    // (a multiplicand the compiler knows to be the high half of a 64-bit value makes its multiply-add a 64-bit
    // multiplication: the halves are opaque too)
    asm("" : "+v"(lo));
    asm("" : "+v"(hi));
#else
    lo = (alo << SH) + blo;
    hi = (ahi << SH) + bhi;
#endif
}

#if defined(GF_CHECKED)
// Host-side checker accumulator: 128-bit, aborts if a 64-bit accumulator would
// have overflowed or gone negative.  It also counts the multiply-accumulates (one v_mad_u64_u32 each on
// the device): bench.py's MACs-per-operation figures come from this counter (the checker build's tests).
inline unsigned long long &gf_mac_counter() {
    static thread_local unsigned long long count = 0;
    return count;
}
struct acc_t {
    unsigned __int128 x;   // the exact value; the device accumulator holds it mod 2^64
    GD_MFN acc_t() : x(0) {}
    GD_MFN explicit acc_t(uint64_t v) : x(v) {}
    // a finished column is read out: it must be what a wrapping u64 accumulator holds
    GD_MFN void chk() const { if (x >> 64) __builtin_trap(); }
    GD_MFN void mac(uint32_t a, uint32_t b) {
        x += (unsigned __int128)a * b;
        gf_mac_counter()++;
    }
    GD_MFN void add(const acc_t &o) { x += o.x; }
    GD_MFN void add_doubled(const acc_t &o) { x += o.x << 1; }
    GD_MFN void add32(uint32_t o) { x += o; }
    GD_MFN void sub(const acc_t &o) { if (o.x > x) __builtin_trap(); x -= o.x; }
    GD_MFN uint32_t lo28() const { chk(); return (uint32_t)x & M28; }
    GD_MFN void shr28() { chk(); x >>= 28; }
    GD_MFN uint32_t lo32() const { if (x >> 32) __builtin_trap(); return (uint32_t)x; }
};
#else
struct acc_t {
    uint64_t x;
    GD_MFN acc_t() : x(0) {}
    GD_MFN explicit acc_t(uint64_t v) : x(v) {}
    // v_mad_u64_u32 acc, a, b, acc.  The empty asm pins the accumulation ORDER (no instruction is
    // emitted for it): without it LLVM reassociates carry + sum(products) into sum(products) +
    // carry and pays an extra 64-bit add per chain and column (16 per multiplication); with it the
    // incoming carry is the first MAC's addend.
    GD_MFN void mac(uint32_t a, uint32_t b) {
        x += (uint64_t)a * b;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(x));
#endif
    }
    GD_MFN void add(const acc_t &o) { x += o.x; }
    GD_MFN void add_doubled(const acc_t &o) { x += o.x << 1; }   // one v_lshl_add_u64
    GD_MFN void add32(uint32_t o) { x += o; }
    GD_MFN void sub(const acc_t &o) { x -= o.x; }
    GD_MFN uint32_t lo28() const { return (uint32_t)x & M28; }
    GD_MFN void shr28() { x >>= 28; }
    GD_MFN uint32_t lo32() const { return (uint32_t)x; }
};
#endif

// ... with a compile-time constant pair as the addend: the constant is a scalar register pair (no moves into VGPRs)
template <uint32_t BLO, uint32_t BHI>
GD_FN void fe_pair_add_const(uint32_t &lo, uint32_t &hi, uint32_t alo, uint32_t ahi) {
#if defined(GF_CHECKED)
    if (((uint64_t)alo + BLO) >> 32 || ((uint64_t)ahi + BHI) >> 32) __builtin_trap();
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GD_NO_PAIRED_ADDS)
    const uint64_t a = (uint64_t)alo | ((uint64_t)ahi << 32);
    const uint64_t b = (uint64_t)BLO | ((uint64_t)BHI << 32);
    uint64_t c;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(c) : "v"(a), "s"(b));
    lo = (uint32_t)c;
    hi = (uint32_t)(c >> 32);
    asm("" : "+v"(lo));
    asm("" : "+v"(hi));
#else
    lo = alo + BLO;
    hi = ahi + BHI;
#endif
}

// ---------------------------------------------------------------- constants

GD_FN fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.v[i] = 0;
    return r;
}
GD_FN fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}
GD_FN fe fe_small(uint32_t w) {  // w < 2^28
    fe r = fe_zero();
    r.v[0] = w;
    return r;
}

// ---------------------------------------------------------------- linear ops

// c = a + b, limb-wise, no carry (mag adds up).
GD_FN fe fe_add(const fe &a, const fe &b) {
    fe c;
#pragma unroll
    for (int k = 0; k < 8; k++) fe_pair_add<0>(c.v[2 * k], c.v[2 * k + 1], a.v[2 * k], a.v[2 * k + 1], b.v[2 * k], b.v[2 * k + 1]);
    return c;
}

// c = a - b + K*p, limb-wise, no carry.  Needs every limb of b <= K*(2^28-1) - K
// (p's limbs are 2^28-1, limb 8 is 2^28-2; cf. gf_bias, arch_x86_64/f_impl.h:36-57).
template <int K>
GD_FN fe fe_sub(const fe &a, const fe &b) {
    fe c;
    constexpr uint32_t B = (uint32_t)K * M28;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#if defined(GF_CHECKED)
        if (b.v[2 * k] > (k == 4 ? B - K : B) || b.v[2 * k + 1] > B) __builtin_trap();
#endif
        // a + K p pair-wise (the bias is a 64-bit constant: one instruction for two limbs), then - b limb by limb
        uint32_t lo, hi;
        if (k == 4) fe_pair_add_const<B - (uint32_t)K, B>(lo, hi, a.v[2 * k], a.v[2 * k + 1]);
        else fe_pair_add_const<B, B>(lo, hi, a.v[2 * k], a.v[2 * k + 1]);
        c.v[2 * k] = lo - b.v[2 * k];
        c.v[2 * k + 1] = hi - b.v[2 * k + 1];
    }
    return c;
}

// One parallel carry pass: limb i keeps its low 28 bits and receives limb i-1's
// overflow; limb 15's overflow re-enters at limbs 0 and 8 (2^448 = 2^224 + 1).
// Result: every limb <= 2^28 - 1 + 15  (cf. gf_weak_reduce, arch_ref64/f_impl.h:30-38).
GD_FN fe fe_weak(const fe &a) {
    fe c;
    const uint32_t top = a.v[15] >> 28;
#pragma unroll
    for (int k = 0; k < 8; k++)   // masked pair + carry pair: five instructions for two limbs
        fe_pair_add<0>(c.v[2 * k], c.v[2 * k + 1], a.v[2 * k] & M28, a.v[2 * k + 1] & M28, k ? a.v[2 * k - 1] >> 28 : top,
                       a.v[2 * k] >> 28);
    c.v[8] += top;
    return c;
}

GD_FN fe fe_select(const fe &a, const fe &b, bool pick_b) {  // pick_b ? b : a
    fe c;
#pragma unroll
    for (int i = 0; i < 16; i++) c.v[i] = pick_b ? b.v[i] : a.v[i];
    return c;
}

// ---------------------------------------------------------------- multiply

// Shared tail of mul/sqr/mulw: lo/hi hold the carries out of limbs 7 and 15.
GD_FN void fe_fold_tails(fe &c, acc_t lo, acc_t hi) {
    lo.add(hi);          // limb 8 receives limb 7's carry and limb 15's (phi^2 = phi + 1)
    lo.add32(c.v[8]);
    hi.add32(c.v[0]);    // limb 0 receives limb 15's carry
    c.v[8] = lo.lo28();
    c.v[0] = hi.lo28();
    lo.shr28();
    hi.shr28();
    c.v[9] += lo.lo32();
    c.v[1] += hi.lo32();
}

// c = a * b mod p.  192 MACs.
GD_FN fe fe_mul(const fe &a, const fe &b) {
    uint32_t sa[8], sb[8], sbb[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        fe_pair_add<0>(sa[j], sa[j + 1], a.v[j], a.v[j + 1], a.v[j + 8], a.v[j + 9]);
        fe_pair_add<0>(sb[j], sb[j + 1], b.v[j], b.v[j + 1], b.v[j + 8], b.v[j + 9]);
        fe_pair_add<1>(sbb[j], sbb[j + 1], b.v[j + 8], b.v[j + 9], b.v[j], b.v[j + 1]);     // b0 + 2 b1
    }
    fe c;
    acc_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc_t cross;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j <= i) {
                cross.mac(a.v[j], b.v[i - j]);          // a0*b0, column i
                hi.mac(sa[j], sb[i - j]);               // (a0+a1)(b0+b1), column i
                lo.mac(a.v[j + 8], b.v[i - j + 8]);     // a1*b1, column i
            } else {                                    // column i+8: one more factor phi
                cross.mac(a.v[j], b.v[i - j + 16]);     // a0*b1
                hi.mac(sa[j], sbb[i - j + 8]);          // (a0+a1)(b0+2*b1)
                lo.mac(a.v[j + 8], sb[i - j + 8]);      // a1*(b0+b1)
            }
        }
        hi.sub(cross);
        lo.add(cross);
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    fe_fold_tails(c, lo, hi);
    return c;
}

// c = a^2 mod p.  136 MACs and no 64-bit combine arithmetic.  With a = a0 + a1 phi,
// s = a0 + a1 and t = 2 a0 + a1:
//     a^2 = (a0^2 + a1^2) + phi * (a1 * t)                       (phi^2 = phi + 1)
// and since a0^2 + a1^2 + a1 t = s^2 + a1^2, the wrapped columns (X_i' = column i+8) give
//     low_i  = (a0^2)_i + (a1^2)_i + (a1 t)_i'        high_i = (a1 t)_i + (s^2)_i' + (a1^2)_i'
// Every term is positive, so each output limb is ONE sum of products whose first addend is the carry
// (a Karatsuba square has 108 MACs but pays 7 64-bit add/sub instructions per column pair).
// A column of a square is 2 * sum_{j<k} x_j x_k + (x_j^2 on the diagonal): the cross products of a limb go
// to their own accumulator and are doubled ONCE, when the column is finished (one v_lshl_add_u64), instead
// of doubling the operands beforehand (24 shifts per square): 4 instructions and 25 hazard no-ops fewer,
// 941 -> 916 SIMD cycles per squaring at two waves per SIMD (tools/fieldbench, sqr_opdbl vs sqr).
// Column bound: high_0 = 3 + 7*(4+1) = 38 products of mag^2 (the multiplication has 46).
template <int COL>
GD_FN void sq_col(acc_t &cross, acc_t &rest, const uint32_t (&x)[8]) {  // column COL (0..14) of x^2
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7 || j > k) continue;
        if (j == k) rest.mac(x[j], x[j]);
        else cross.mac(x[j], x[k]);
    }
}
template <int COL>
GD_FN void mul_col(acc_t &acc, const uint32_t (&x)[8], const uint32_t (&y)[8]) {  // column COL of x*y
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7) continue;
        acc.mac(x[j], y[k]);
    }
}
template <int I>
GD_FN void sqr_column(fe &c, acc_t &lo, acc_t &hi, const uint32_t (&u)[8], const uint32_t (&v)[8],
                      const uint32_t (&s)[8], const uint32_t (&t)[8]) {
    acc_t lo_cross, hi_cross;
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
GD_FN fe fe_sqr(const fe &a) {
    uint32_t u[8], v[8], s[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u[j] = a.v[j];
        v[j] = a.v[j + 8];
    }
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        fe_pair_add<0>(s[j], s[j + 1], a.v[j], a.v[j + 1], a.v[j + 8], a.v[j + 9]);
        fe_pair_add<1>(t[j], t[j + 1], a.v[j], a.v[j + 1], a.v[j + 8], a.v[j + 9]);         // 2 a0 + a1
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

// c = a * w, w < 2^32 (cf. gf_mulw_unsigned, arch_ref64/f_impl.c:168-190).  16 MACs.
// Any a with 32-bit limbs is fine; result is mag 1.
GD_FN fe fe_mulw(const fe &a, uint32_t w) {
    fe c;
    acc_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lo.mac(a.v[i], w);
        hi.mac(a.v[i + 8], w);
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    fe_fold_tails(c, lo, hi);
    return c;
}

GD_FN fe fe_sqrn(fe x, int n) {
    for (int i = 0; i < n; i++) x = fe_sqr(x);
    return x;
}

// ---------------------------------------------------------------- canonical form

// Canonical representative in [0, p): limbs < 2^28  (cf. gf_strong_reduce,
// src/f_generic.c:71-105).  Input: any mag <= 15.
GD_FN fe fe_strong(const fe &a) {
    fe c = fe_weak(fe_weak(a));  // now value < 2p and limbs <= 2^28
    // subtract p with a signed ripple
    int64_t sc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        sc += (int64_t)c.v[i] - (int64_t)(i == 8 ? M28 - 1 : M28);
        c.v[i] = (uint32_t)sc & M28;
        sc >>= 28;
    }
    // sc == 0: value was >= p, done.  sc == -1: went negative, add p back.
    uint32_t addback = (uint32_t)sc;  // 0 or 0xffffffff
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        carry += (uint64_t)c.v[i] + (addback & (i == 8 ? M28 - 1 : M28));
        c.v[i] = (uint32_t)carry & M28;
        carry >>= 28;
    }
    return c;
}

GD_FN bool fe_is_zero(const fe &a) {  // a == 0 mod p
    fe c = fe_strong(a);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc |= c.v[i];
    return acc == 0;
}
GD_FN bool fe_eq(const fe &a, const fe &b) {  // a, b mag <= 2
    return fe_is_zero(fe_sub<2>(a, fe_weak(b)));
}
GD_FN bool fe_lobit(const fe &a) { return fe_strong(a).v[0] & 1; }

GD_FN fe fe_neg(const fe &a) { return fe_sub<2>(fe_zero(), fe_weak(a)); }  // mag 2
GD_FN fe fe_cond_neg(const fe &a, bool neg) { return fe_select(a, fe_neg(a), neg); }

// x^((p-3)/4) = 1/sqrt(x) up to sign (cf. gf_isr, src/f_arithmetic.c:14-46).
// Chain over x^(2^k - 1), k = 1,2,3,6,9,18,19,37,74,111,222,223.  Input mag <= 2.
// *ok = (result^2 * x == 1).
GD_FN fe fe_isr(const fe &x, bool *ok) {
    fe e1 = fe_weak(x);
    fe e2 = fe_mul(fe_sqr(e1), e1);
    fe e3 = fe_mul(fe_sqr(e2), e1);
    fe e6 = fe_mul(fe_sqrn(e3, 3), e3);
    fe e9 = fe_mul(fe_sqrn(e6, 3), e3);
    fe e18 = fe_mul(fe_sqrn(e9, 9), e9);
    fe e19 = fe_mul(fe_sqr(e18), e1);
    fe e37 = fe_mul(fe_sqrn(e19, 18), e18);
    fe e74 = fe_mul(fe_sqrn(e37, 37), e37);
    fe e111 = fe_mul(fe_sqrn(e74, 37), e37);
    fe e222 = fe_mul(fe_sqrn(e111, 111), e111);
    fe e223 = fe_mul(fe_sqr(e222), e1);
    fe r = fe_mul(fe_sqrn(e223, 223), e222);
    fe chk = fe_mul(fe_sqr(r), e1);
    *ok = fe_eq(chk, fe_one());
    return r;
}

// 1/x (0 -> 0)  (cf. gf_invert, src/goldilocks.c:69-80)
GD_FN fe fe_invert(const fe &x) {
    bool ok;
    fe xr = fe_weak(x);
    fe t = fe_isr(fe_sqr(xr), &ok);
    return fe_mul(fe_sqr(t), xr);
}

// ---------------------------------------------------------------- ABI conversion

// From the reference ABI: 8 x u64 limbs of 56 bits each (weakly reduced inputs
// may exceed 2^56 slightly; anything up to 2^60 per limb is absorbed here).
GD_FN fe fe_from_limbs56(const uint64_t l[8]) {
    fe c;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c.v[2 * i] = (uint32_t)l[i] & M28;
        c.v[2 * i + 1] = (uint32_t)(l[i] >> 28);  // keeps the excess above 2^56 (<= 4 bits)
    }
    return c;
}
// To the reference ABI, weakly reduced (each 56-bit limb < 2^56 + 2^33).
GD_FN void fe_to_limbs56(uint64_t l[8], const fe &a) {
    fe c = fe_weak(a);
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = (uint64_t)c.v[2 * i] + ((uint64_t)c.v[2 * i + 1] << 28);
}

// 56-byte little-endian wire format of the canonical value as 14 x u32
// (cf. gf_serialize, src/f_generic.c:19-38).
GD_FN void fe_serialize_words(uint32_t w[14], const fe &a) {
    fe c = fe_strong(a);
    // limb i occupies bits [28i, 28i+28)
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int lo_bit = 28 * i - 32 * k;  // position of limb i relative to word k
            if (lo_bit > -28 && lo_bit < 32) {
                if (lo_bit >= 0) acc |= (uint64_t)c.v[i] << lo_bit;
                else acc |= (uint64_t)c.v[i] >> (-lo_bit);
            }
        }
        w[k] = (uint32_t)acc;
    }
}
// 14 words (16 x 28 bits, little-endian bit string) -> limbs, no range check
GD_FN fe fe_unpack_words(const uint32_t w[14]) {
    fe c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int bit = 28 * i, k = bit / 32, sh = bit % 32;
        uint64_t two = (uint64_t)w[k] | (k + 1 < 14 ? (uint64_t)w[k + 1] << 32 : 0);
        c.v[i] = (uint32_t)(two >> sh) & M28;
    }
    return c;
}
// Returns false iff the 448-bit value read is >= p (cf. gf_deserialize,
// src/f_generic.c:49-68).  The result limbs are < 2^28 either way.
GD_FN bool fe_deserialize_words(fe &c, const uint32_t w[14]) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int bit = 28 * i, k = bit / 32, sh = bit % 32;
        uint64_t two = (uint64_t)w[k] | (k + 1 < 14 ? (uint64_t)w[k + 1] << 32 : 0);
        c.v[i] = (uint32_t)(two >> sh) & M28;
    }
    int64_t sc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sc = (sc + (int64_t)c.v[i] - (int64_t)(i == 8 ? M28 - 1 : M28)) >> 28;
    return sc != 0;  // negative => value < p
}

}  // namespace gd
