// sc14.hpp -- scalars mod q (q = group order, 446 bits) as 14 x u32 words, one scalar
// per lane.  Restates what the hot path needs from the reference's src/scalar.c:
// add (:176-189), sub (:168-174), halve (:316-332), product and decode_long (:55-100, :257-293: by folding at
// the modulus' size here instead of Montgomery products -- same values).  None of this is performance critical (< 0.1 % of a
// scalarmul), so it is written for clarity with 64-bit carries.
#pragma once
#include "gf28.hpp"

namespace gd {

struct sc {
    uint32_t w[14];
};

// q, little-endian 32-bit words (src/scalar.c:18-20)
GD_CONST uint32_t SC_Q[14] = {0xab5844f3u, 0x2378c292u, 0x8dc58f55u, 0x216cc272u, 0xaed63690u,
                              0xc44edb49u, 0x7cca23e9u, 0xffffffffu, 0xffffffffu, 0xffffffffu,
                              0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
// (2^450 - 1) mod q: signed-window recoding offset (src/goldilocks.c:33-37)
GD_CONST uint32_t SC_ADJ[14] = {0x4a7bb0cfu, 0xc873d6d5u, 0x23a70aadu, 0xe933d8d7u, 0x129c96fdu,
                                0xbb124b65u, 0x335dc163u, 0x00000008u, 0, 0, 0, 0, 0, 0};
// (2^448 - 1) mod q: the same recoding for 56 signed 8-bit windows (fixed-base window table)
GD_CONST uint32_t SC_ADJ8[14] = {0x529eec33u, 0x721cf5b5u, 0xc8e9c2abu, 0x7a4cf635u, 0x44a725bfu,
                                 0xeec492d9u, 0x0cd77058u, 0x00000002u, 0, 0, 0, 0, 0, 0};
// (2^456 - 1) mod q: the same for 38 signed 12-bit windows (experiment, -DGD_BWT_BITS=12)
GD_CONST uint32_t SC_ADJ12[14] = {0x9eec33ffu, 0x1cf5b552u, 0xe9c2ab72u, 0x4cf635c8u, 0xa725bf7au,
                                  0xc492d944u, 0xd77058eeu, 0x0000020cu, 0, 0, 0, 0, 0, 0};
// (2^460 - 1) mod q and (2^462 - 1) mod q: 23 signed 20-bit and 21 signed 22-bit windows (-DGD_BWT_BITS=20 / 22)
GD_CONST uint32_t SC_ADJ20[14] = {0xeec33fffu, 0xcf5b5529u, 0x9c2ab721u, 0xcf635c8eu, 0x725bf7a4u,
                                  0x492d944au, 0x77058eecu, 0x000020cdu, 0, 0, 0, 0, 0, 0};
GD_CONST uint32_t SC_ADJ22[14] = {0xbb0cffffu, 0x3d6d54a7u, 0x70aadc87u, 0x3d8d723au, 0xc96fde93u,
                                  0x24b65129u, 0xdc163bb1u, 0x00008335u, 0, 0, 0, 0, 0, 0};

GD_FN sc sc_zero() {
    sc r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.w[i] = 0;
    return r;
}
GD_FN sc sc_const(const uint32_t *k) {
    sc r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.w[i] = k[i];
    return r;
}

// out = {extra, a} - b, then + q if that is negative (src/scalar.c:30-53).
GD_FN sc sc_subx(const sc &a, const sc &b, uint32_t extra) {
    sc o;
    int64_t chain = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        chain += (int64_t)a.w[i] - (int64_t)b.w[i];
        o.w[i] = (uint32_t)chain;
        chain >>= 32;
    }
    uint32_t borrow = (uint32_t)chain + extra;  // 0 or 0xffffffff
    uint64_t c2 = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        c2 += (uint64_t)o.w[i] + (SC_Q[i] & borrow);
        o.w[i] = (uint32_t)c2;
        c2 >>= 32;
    }
    return o;
}
GD_FN sc sc_sub(const sc &a, const sc &b) { return sc_subx(a, b, 0); }
GD_FN sc sc_add(const sc &a, const sc &b) {
    sc t;
    uint64_t chain = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        chain += (uint64_t)a.w[i] + b.w[i];
        t.w[i] = (uint32_t)chain;
        chain >>= 32;
    }
    return sc_subx(t, sc_const(SC_Q), (uint32_t)chain);
}
GD_FN sc sc_halve(const sc &a) {
    uint32_t mask = 0u - (a.w[0] & 1);
    sc t;
    uint64_t chain = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        chain += (uint64_t)a.w[i] + (SC_Q[i] & mask);
        t.w[i] = (uint32_t)chain;
        chain >>= 32;
    }
    sc o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.w[i] = t.w[i] >> 1 | t.w[i + 1] << 31;
    o.w[13] = t.w[13] >> 1 | (uint32_t)chain << 31;
    return o;
}

// Reduction mod q by folding at the modulus' own size: q = 2^446 - c with c of 224 bits, so
// x = lo + 2^446 hi == lo + c hi (mod q).  (The reference multiplies and reduces with Montgomery products,
// src/scalar.c:55-100; only values are observable, and these are the same canonical ones.)
GD_CONST uint32_t SC_C[7] = {0x54a7bb0du, 0xdc873d6du, 0x723a70aau, 0xde933d8du, 0x5129c96fu, 0x3bb124b6u, 0x8335dc16u};
// out[NIN' = 14 + ...]: x[NIN] folded once.  NHI = words of x >> 446 that can be nonzero.
template <int NIN, int NHI, int NOUT>
GD_FN void sc_fold(uint32_t (&out)[NOUT], const uint32_t (&x)[NIN]) {
    static_assert(NOUT >= NHI + 8 && NOUT >= 15 && NIN >= 13 + NHI, "sizes");
    uint32_t xp[NIN + 1], hi[NHI];    // (one zero word of padding: the last shifted read needs no bounds test)
#pragma unroll
    for (int k = 0; k <= NIN; k++) xp[k] = k < NIN ? x[k] : 0u;
#pragma unroll
    for (int k = 0; k < NHI; k++) hi[k] = xp[13 + k] >> 30 | xp[14 + k] << 2;
#pragma unroll
    for (int k = 0; k < NOUT; k++) out[k] = k < 13 ? x[k] : k == 13 ? x[13] & 0x3fffffffu : 0u;
#pragma unroll
    for (int i = 0; i < NHI; i++) {     // out += c * hi[i] << (32 i)
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            carry += (uint64_t)hi[i] * SC_C[j] + out[i + j];
            out[i + j] = (uint32_t)carry;
            carry >>= 32;
        }
#pragma unroll
        for (int k = i + 7; k < NOUT; k++) {   // (only the first of these can see a nonzero carry twice in a row)
            carry += out[k];
            out[k] = (uint32_t)carry;
            carry >>= 32;
        }
    }
}
// x (15 words, below 2 q) -> canonical
GD_FN sc sc_final(const uint32_t (&y)[15]) {
    sc t;
#pragma unroll
    for (int i = 0; i < 14; i++) t.w[i] = y[i];
    return sc_subx(t, sc_const(SC_Q), 0);
}
// a*b mod q: schoolbook (196 word products), then 896 -> 675 -> 454 -> 447 bits in three folds
GD_FN sc sc_mul(const sc &a, const sc &b) {
    uint32_t x[28];
#pragma unroll
    for (int i = 0; i < 28; i++) x[i] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 14; j++) {
            carry += (uint64_t)a.w[i] * b.w[j] + x[i + j];
            x[i + j] = (uint32_t)carry;
            carry >>= 32;
        }
        x[i + 14] = (uint32_t)carry;
    }
    uint32_t y1[23], y2[16], y[15];
    sc_fold<28, 15, 23>(y1, x);
    sc_fold<23, 8, 16>(y2, y1);
    sc_fold<16, 1, 15>(y, y2);
    return sc_final(y);
}

// Reduce an arbitrary 448-bit word string mod q ("ham-handed reduce", scalar.c:246): one fold
GD_FN sc sc_reduce(const sc &a) {
    uint32_t x[15], y[15];
#pragma unroll
    for (int i = 0; i < 15; i++) x[i] = i < 14 ? a.w[i] : 0u;
    sc_fold<15, 1, 15>(y, x);
    return sc_final(y);
}

// s' = (s + 2^450 - 1)/2 mod q: recoding for signed fixed windows
// (src/goldilocks.c:420-421 and :842-843).
GD_FN sc sc_recode_signed(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ))); }

// W = (s + 2^448 - 1)/2 mod q: its 56 bytes w_i encode s = sum (2 w_i - 255) 256^i, odd digits in
// [-255, 255] (same derivation as above with 8-bit windows: sum 255*256^i = 2^448 - 1).
GD_FN sc sc_recode_signed8(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ8))); }
GD_FN sc sc_recode_signed12(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ12))); }
GD_FN sc sc_recode_signed20(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ20))); }
GD_FN sc sc_recode_signed22(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ22))); }

}  // namespace gd
