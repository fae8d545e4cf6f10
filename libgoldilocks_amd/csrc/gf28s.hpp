// gf28s.hpp -- GF(2^448 - 2^224 - 1) with SIGNED 28-bit limbs held as 8 register PAIRS: the field layer of the
// ladders (montgomery.hpp, x448.hpp), where every kernel is bound by the NUMBER of VALU instructions it issues
// (one instruction per SIMD every 4 cycles whatever its kind).
//
// Two things the unsigned layer (gf28.hpp) pays for are not paid here:
//   * a difference is a - b, limb by limb, and may be negative: no bias K*p is added (16 instructions per
//     subtraction) and no weak reduction has to bring the biased difference (mag 3) back under the multiplier's
//     limit (48 instructions) -- a difference of two products is as small as the products.  The products are
//     v_mad_i64_i32, the carries arithmetic shifts; a product's limbs are masked, so they come out in
//     [0, 2^28) whatever the signs that went in (limbs 1 and 9 receive the last carry and may leave that range
//     by a few units either way).
//   * limbs 2k and 2k+1 share a 64-bit register pair w[k] (low half, high half).  Where the EVEN limbs of both
//     operands are known to be non-negative and their sum to stay below 2^32 -- products, and sums of products --
//     an addition is ONE v_lshl_add_u64 per pair (8 per element instead of 16): the low half cannot carry into the
//     high half, and a negative HIGH half wraps inside its own 32 bits like any two's complement number.  The
//     type carries that knowledge: sfe<true> ("pairable") or sfe<false> ("signed": a difference, or a sum with one).
//
// Same identity as gf28.hpp's fe_mul / fe_sqr (restating src/arch_ref64/f_impl.c:7-166, :192-300; math in SURVEY.md
// section 9): phi = 2^224, phi^2 = phi + 1, three 8x8-limb half products.
//
// MAGNITUDE CONTRACT (checked in the host-side checker build, GF_CHECKED, tests/hostsim):
//   |limb| <= m * 2^28 (+ a few units) is "mag m".  mul/sqr results are mag 1 and pairable.  A finished column of a
//   product is at most 38 limb products: 38 * maxlimb(a) * maxlimb(b) < 2^63, i.e. mag(a) * mag(b) <= 3.3
//   (sum x difference = 2 x 1 is fine, sum x sum = 4 is not), and the pre-added halves (a0 + a1, b0 + 2 b1) have
//   to fit 31 bits: mag(a) <= 3, mag(b) <= 2.  sqr(a): mag(a) <= 1.8.  Pairable additions need the even limbs of
//   both operands in [0, 2^31).
#pragma once
#include "gf28.hpp"

namespace gd {

#if defined(GF_CHECKED)
struct sacc_t {
    __int128 x;
    GD_MFN sacc_t() : x(0) {}
    GD_MFN explicit sacc_t(int64_t v) : x(v) {}
    GD_MFN void chk() const { if (x >= ((__int128)1 << 63) || x < -((__int128)1 << 63)) __builtin_trap(); }
    GD_MFN void mac(int32_t a, int32_t b) {
        x += (__int128)a * b;
        gf_mac_counter()++;
    }
    GD_MFN void add(const sacc_t &o) { x += o.x; }
    GD_MFN void add_doubled(const sacc_t &o) { x += o.x * 2; }
    GD_MFN void add32(int32_t o) { x += o; }
    GD_MFN void sub(const sacc_t &o) { x -= o.x; }
    // a finished column is read out (lo28, then one of the shifts): it must be what the wrapping 64-bit accumulator
    // holds, read as a signed number (shr28) or as an unsigned one (shr28_u)
    GD_MFN int32_t lo28() const { return (int32_t)((uint32_t)(uint64_t)x & M28); }
    GD_MFN void shr28() { chk(); x >>= 28; }
    GD_MFN void shr28_u() { if (x < 0 || (x >> 64)) __builtin_trap(); x >>= 28; }
    GD_MFN int32_t lo32() const { if (x >= ((__int128)1 << 31) || x < -((__int128)1 << 31)) __builtin_trap(); return (int32_t)x; }
};
#else
struct sacc_t {
    int64_t x;
    GD_MFN sacc_t() : x(0) {}
    GD_MFN explicit sacc_t(int64_t v) : x(v) {}
    // v_mad_i64_i32 acc, a, b, acc; the empty asm pins the accumulation order (gf28.hpp acc_t::mac)
    // (every sum is taken modulo 2^64: the three high-half columns that are read as unsigned numbers offset by 2^62 go
    // past INT64_MAX on purpose, which a signed `+=` would make undefined)
    GD_MFN void mac(int32_t a, int32_t b) {
        x = (int64_t)((uint64_t)x + (uint64_t)((int64_t)a * b));
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(x));
#endif
    }
    GD_MFN void add(const sacc_t &o) { x = (int64_t)((uint64_t)x + (uint64_t)o.x); }
    GD_MFN void add_doubled(const sacc_t &o) { x = (int64_t)(((uint64_t)o.x << 1) + (uint64_t)x); }   // one v_lshl_add_u64
    GD_MFN void add32(int32_t o) { x = (int64_t)((uint64_t)x + (uint64_t)(int64_t)o); }
    GD_MFN void sub(const sacc_t &o) { x = (int64_t)((uint64_t)x - (uint64_t)o.x); }
    // The mask's result is hidden from the compiler's known-bits analysis: a multiplicand it can prove non-negative
    // is zero-extended, and zext * sext is not a v_mad_i64_i32 but an expansion into two multiply-adds and two moves
    // (tools/fieldbench dbl_signed, round 2: 231 v_mov_b32 per doubling).
    GD_MFN int32_t lo28() const {
        int32_t r = (int32_t)((uint32_t)x & M28);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(r));
#endif
        return r;
    }
    GD_MFN void shr28() { x >>= 28; }
    GD_MFN void shr28_u() { x = (int64_t)((uint64_t)x >> 28); }
    GD_MFN int32_t lo32() const { return (int32_t)x; }
};
#endif

// PAIRABLE: the even limbs are known to be in [0, 2^31) and to stay there when two such elements are added
template <bool PAIRABLE>
struct sfe {
    int32_t v[16];   // limbs 2k and 2k+1 are meant to share an aligned register pair (the pair-wise additions ask for it)
};
using sfp = sfe<true>;
using sfs = sfe<false>;

GD_FN int32_t s_check32(int64_t v) {
#if defined(GF_CHECKED)
    if (v >= (1ll << 31) || v < -(1ll << 31)) __builtin_trap();
#endif
    return (int32_t)v;
}

// (a << SH) + b on the limb pair (lo, hi): one v_lshl_add_u64 when both are pairable, two 32-bit instructions otherwise
template <bool PAIRED, int SH>
GD_FN void s_pair_add(int32_t &lo, int32_t &hi, int32_t alo, int32_t ahi, int32_t blo, int32_t bhi) {
    if (PAIRED) {
#if defined(GF_CHECKED)
        // even limbs non-negative, and their sum (a's doubled when SH) stays one: nothing reaches the high half
        if (alo < 0 || blo < 0 || ((int64_t)alo << SH) + blo >= (1ll << 31)) __builtin_trap();
        s_check32(((int64_t)ahi << SH) + bhi);
#endif
        const uint64_t a = (uint64_t)(uint32_t)alo | ((uint64_t)(uint32_t)ahi << 32);
        const uint64_t b = (uint64_t)(uint32_t)blo | ((uint64_t)(uint32_t)bhi << 32);
#if defined(__HIP_DEVICE_COMPILE__)
        // Written out: left to itself the compiler takes the 64-bit addition apart again (pair + zext(low half), a
        // 32-bit addition of the high half, a move for the zero), and a multiplicand it knows to be the high half of a
        // 64-bit value turns its multiply-add into a full 64-bit multiplication -- so the halves are opaque as well.
        uint64_t c;
        asm("v_lshl_add_u64 %0, %1, %3, %2" : "=v"(c) : "v"(a), "v"(b), "n"(SH));
        lo = (int32_t)(uint32_t)c;
        hi = (int32_t)(uint32_t)(c >> 32);
        asm("" : "+v"(lo));
        asm("" : "+v"(hi));
#else
        const uint64_t c = (a << SH) + b;
        lo = (int32_t)(uint32_t)c;
        hi = (int32_t)(uint32_t)(c >> 32);
#endif
    } else {
        lo = s_check32(((int64_t)alo << SH) + blo);
        hi = s_check32(((int64_t)ahi << SH) + bhi);
    }
}

// ---------------------------------------------------------------- conversions

// from the unsigned layer (any mag whose limbs fit 31 bits): pairable as it stands
GD_FN sfp sfe_from_fe(const fe &a) {
    sfp c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        c.v[i] = (int32_t)a.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(c.v[i]));   // opaque, like a product's limbs (sacc_t::lo28): no zero-extended multiplicands
#endif
    }
    return c;
}
// to the unsigned layer, mag 1: limb-wise + 2p (mag <= 2 in, so nothing is negative), then one carry pass.
template <bool P>
GD_FN fe sfe_to_fe(const sfe<P> &a) {
    fe c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int32_t v = a.v[i] + (int32_t)(i == 8 ? 2 * M28 - 2 : 2 * M28);
#if defined(GF_CHECKED)
        if (v < 0) __builtin_trap();
#endif
        c.v[i] = (uint32_t)v;
    }
    return fe_weak(c);
}

// ---------------------------------------------------------------- linear ops

template <bool PA, bool PB>
GD_FN sfe<PA && PB> sfe_add(const sfe<PA> &a, const sfe<PB> &b) {
    sfe<PA && PB> c;
#pragma unroll
    for (int k = 0; k < 8; k++)
        s_pair_add<PA && PB, 0>(c.v[2 * k], c.v[2 * k + 1], a.v[2 * k], a.v[2 * k + 1], b.v[2 * k], b.v[2 * k + 1]);
    return c;
}
template <bool PA, bool PB>
GD_FN sfs sfe_sub(const sfe<PA> &a, const sfe<PB> &b) {
    sfs c;
#pragma unroll
    for (int i = 0; i < 16; i++) c.v[i] = s_check32((int64_t)a.v[i] - b.v[i]);
    return c;
}
template <bool P>
GD_FN sfe<P> sfe_select(const sfe<P> &a, const sfe<P> &b, bool pick_b) {  // pick_b ? b : a
    sfe<P> c;
#pragma unroll
    for (int i = 0; i < 16; i++) c.v[i] = pick_b ? b.v[i] : a.v[i];
    return c;
}

// ---------------------------------------------------------------- multiply

struct shalf {   // the 8 limbs of a half (or of a pre-added pair of halves) as multiplicands
    int32_t v[8];
};
GD_FN shalf s_half(const int32_t *v) {
    shalf h;
#pragma unroll
    for (int j = 0; j < 8; j++) h.v[j] = v[j];
    return h;
}
template <bool P, int SH>
GD_FN shalf s_half_sum(const int32_t *a, const int32_t *b) {   // (a << SH) + b
    shalf h;
#pragma unroll
    for (int k = 0; k < 4; k++)
        s_pair_add<P, SH>(h.v[2 * k], h.v[2 * k + 1], a[2 * k], a[2 * k + 1], b[2 * k], b[2 * k + 1]);
    return h;
}

// the multiplier's side of a product, pre-added once: what a loop-invariant factor keeps (montgomery.hpp's x1)
struct smultiplier {
    shalf b0, b1, sb, sbb;   // b0, b1, b0 + b1, b0 + 2 b1
};
template <bool P>
GD_FN smultiplier s_multiplier(const sfe<P> &b) {
    smultiplier m;
    m.b0 = s_half(b.v);
    m.b1 = s_half(b.v + 8);
    m.sb = s_half_sum<P, 0>(b.v, b.v + 8);
    m.sbb = s_half_sum<P, 1>(b.v + 8, b.v);
    return m;
}

using sfe_builder = sfp;   // the limbs of a result, as they are produced

GD_FN void s_fold_tails(sfe_builder &c, sacc_t lo, sacc_t hi) {
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
template <bool PA>
GD_FN sfp sfe_mul(const sfe<PA> &a, const smultiplier &b) {
    const shalf a0 = s_half(a.v), a1 = s_half(a.v + 8), sa = s_half_sum<PA, 0>(a.v, a.v + 8);
    sfe_builder c;
    sacc_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        sacc_t cross;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j <= i) {
                cross.mac(a0.v[j], b.b0.v[i - j]);          // a0*b0, column i
                hi.mac(sa.v[j], b.sb.v[i - j]);             // (a0+a1)(b0+b1), column i
                lo.mac(a1.v[j], b.b1.v[i - j]);             // a1*b1, column i
            } else {                                        // column i+8: one more factor phi
                cross.mac(a0.v[j], b.b1.v[i - j + 8]);      // a0*b1
                hi.mac(sa.v[j], b.sbb.v[i - j + 8]);        // (a0+a1)(b0+2*b1)
                lo.mac(a1.v[j], b.sb.v[i - j + 8]);         // a1*(b0+b1)
            }
        }
        hi.sub(cross);
        lo.add(cross);
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    s_fold_tails(c, lo, hi);
    return c;
}
template <bool PA, bool PB>
GD_FN sfp sfe_mul(const sfe<PA> &a, const sfe<PB> &b) {
    return sfe_mul(a, s_multiplier(b));
}

// c = a^2 mod p.  136 MACs (gf28.hpp fe_sqr: every wrapped column a sum of products, cross terms doubled once).
template <int COL>
GD_FN void ssq_col(sacc_t &cross, sacc_t &rest, const shalf &x) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7 || j > k) continue;
        if (j == k) rest.mac(x.v[j], x.v[j]);
        else cross.mac(x.v[j], x.v[k]);
    }
}
template <int COL>
GD_FN void smul_col(sacc_t &acc, const shalf &x, const shalf &y) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = COL - j;
        if (k < 0 || k > 7) continue;
        acc.mac(x.v[j], y.v[k]);
    }
}
// SUM2: the input is a sum of two products (mag 2, every limb >= -2^8).  Columns 0, 1 and 2 of the high half then
// hold up to 38, 36 and 34 products of 2^58: more than a signed 64-bit number (32), less than an unsigned one --
// and they cannot be far below zero.  Those three columns are read as UNSIGNED numbers offset by 2^62: the offset
// is the cross accumulator's start value (half of it, the accumulator is doubled: an operand of its first
// multiply-add, no instruction), the carry out is 2^34 too large, which the next column's start value takes back.
template <int I, bool SUM2>
GD_FN void ssqr_column(sfe_builder &c, sacc_t &lo, sacc_t &hi, const shalf &u, const shalf &v, const shalf &s,
                       const shalf &t) {
    constexpr int64_t HALF_OFFSET = !SUM2 ? 0
                                    : I == 0 ? (1ll << 61)
                                    : I <= 2 ? (1ll << 61) - (1ll << 33)
                                    : I == 3 ? -(1ll << 33) : 0;
    sacc_t lo_cross, hi_cross(HALF_OFFSET);
    ssq_col<I>(lo_cross, lo, u);
    ssq_col<I>(lo_cross, lo, v);
    smul_col<I>(hi, v, t);
    if (I < 7) {
        smul_col<I + 8>(lo, v, t);
        ssq_col<I + 8>(hi_cross, hi, s);
        ssq_col<I + 8>(hi_cross, hi, v);
    }
    if (I > 0) lo.add_doubled(lo_cross);
    if (I < 7) hi.add_doubled(hi_cross);
    c.v[I] = lo.lo28();
    c.v[I + 8] = hi.lo28();
    lo.shr28();
    if (SUM2 && I <= 2) hi.shr28_u();
    else hi.shr28();
}
template <bool SUM2, bool P>
GD_FN sfp sfe_sqr(const sfe<P> &a) {
    static_assert(P || !SUM2, "a sum of two products is pairable");
    const shalf u = s_half(a.v), v = s_half(a.v + 8), s = s_half_sum<P, 0>(a.v, a.v + 8),
                t = s_half_sum<P, 1>(a.v, a.v + 8);                              // 2 a0 + a1
    sfe_builder c;
    sacc_t lo, hi;
    ssqr_column<0, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<1, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<2, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<3, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<4, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<5, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<6, SUM2>(c, lo, hi, u, v, s, t);
    ssqr_column<7, SUM2>(c, lo, hi, u, v, s, t);
    s_fold_tails(c, lo, hi);
    return c;
}

// c = a * w, 0 <= w < 2^31.  16 MACs.
template <bool P>
GD_FN sfp sfe_mulw(const sfe<P> &a, int32_t w) {
    sfe_builder c;
    sacc_t lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lo.mac(a.v[i], w);
        hi.mac(a.v[i + 8], w);
        c.v[i] = lo.lo28();
        c.v[i + 8] = hi.lo28();
        lo.shr28();
        hi.shr28();
    }
    s_fold_tails(c, lo, hi);
    return c;
}

}  // namespace gd
