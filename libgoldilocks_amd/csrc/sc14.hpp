// sc14.hpp -- scalars mod q (q = group order, 446 bits) as 14 x u32 words, one scalar
// per lane.  Restates what the hot path needs from the reference's src/scalar.c:
// add (:176-189), sub (:168-174), halve (:316-332), Montgomery product (:55-91),
// decode_long (:257-293).  None of this is performance critical (< 0.1 % of a
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
// R^2 mod q, R = 2^448 (src/scalar.c:20-22)
GD_CONST uint32_t SC_R2[14] = {0x049b9b60u, 0xe3539257u, 0xc1b195d9u, 0x7af32c4bu, 0x88ea1859u,
                               0x0d66de23u, 0x5ee4d838u, 0xae17cf72u, 0xa3c47c44u, 0x1a9cc14bu,
                               0xe4d070afu, 0x2052bcb7u, 0xf823b729u, 0x3402a939u};
// (2^450 - 1) mod q: signed-window recoding offset (src/goldilocks.c:33-37)
GD_CONST uint32_t SC_ADJ[14] = {0x4a7bb0cfu, 0xc873d6d5u, 0x23a70aadu, 0xe933d8d7u, 0x129c96fdu,
                                0xbb124b65u, 0x335dc163u, 0x00000008u, 0, 0, 0, 0, 0, 0};
// (2^448 - 1) mod q: the same recoding for 56 signed 8-bit windows (fixed-base window table)
GD_CONST uint32_t SC_ADJ8[14] = {0x529eec33u, 0x721cf5b5u, 0xc8e9c2abu, 0x7a4cf635u, 0x44a725bfu,
                                 0xeec492d9u, 0x0cd77058u, 0x00000002u, 0, 0, 0, 0, 0, 0};
// (2^456 - 1) mod q: the same for 38 signed 12-bit windows (experiment, -DGD_BWT_BITS=12)
GD_CONST uint32_t SC_ADJ12[14] = {0x9eec33ffu, 0x1cf5b552u, 0xe9c2ab72u, 0x4cf635c8u, 0xa725bf7au,
                                  0xc492d944u, 0xd77058eeu, 0x0000020cu, 0, 0, 0, 0, 0, 0};
// -q^-1 mod 2^32 (low word of src/scalar.c:17 MONTGOMERY_FACTOR)
constexpr uint32_t SC_MONT32 = 0xae918bc5u;

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

// a*b/2^448 mod q, word-serial Montgomery (src/scalar.c:55-91 with 32-bit words;
// -q^-1 mod 2^32 is the low half of the 64-bit factor, so the result is the same
// residue and, after the final conditional subtraction, the same canonical value).
GD_FN sc sc_montmul(const sc &a, const sc &b) {
    uint32_t acc[15];
#pragma unroll
    for (int i = 0; i < 15; i++) acc[i] = 0;
    uint32_t hi_carry = 0;
    for (int i = 0; i < 14; i++) {
        uint32_t m = a.w[i];
        uint64_t chain = 0;
#pragma unroll
        for (int j = 0; j < 14; j++) {
            chain += (uint64_t)m * b.w[j] + acc[j];
            acc[j] = (uint32_t)chain;
            chain >>= 32;
        }
        acc[14] = (uint32_t)chain;
        m = acc[0] * SC_MONT32;
        chain = 0;
#pragma unroll
        for (int j = 0; j < 14; j++) {
            chain += (uint64_t)m * SC_Q[j] + acc[j];
            if (j) acc[j - 1] = (uint32_t)chain;
            chain >>= 32;
        }
        chain += (uint64_t)acc[14] + hi_carry;
        acc[13] = (uint32_t)chain;
        hi_carry = (uint32_t)(chain >> 32);
    }
    sc t;
#pragma unroll
    for (int i = 0; i < 14; i++) t.w[i] = acc[i];
    return sc_subx(t, sc_const(SC_Q), hi_carry);
}
GD_FN sc sc_mul(const sc &a, const sc &b) { return sc_montmul(sc_montmul(a, b), sc_const(SC_R2)); }

// Reduce an arbitrary 448-bit word string mod q ("ham-handed reduce", scalar.c:246)
GD_FN sc sc_reduce(const sc &a) {
    sc one = sc_zero();
    one.w[0] = 1;
    return sc_mul(a, one);
}

// s' = (s + 2^450 - 1)/2 mod q: recoding for signed fixed windows
// (src/goldilocks.c:420-421 and :842-843).
GD_FN sc sc_recode_signed(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ))); }

// W = (s + 2^448 - 1)/2 mod q: its 56 bytes w_i encode s = sum (2 w_i - 255) 256^i, odd digits in
// [-255, 255] (same derivation as above with 8-bit windows: sum 255*256^i = 2^448 - 1).
GD_FN sc sc_recode_signed8(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ8))); }
GD_FN sc sc_recode_signed12(const sc &s) { return sc_halve(sc_add(s, sc_const(SC_ADJ12))); }

}  // namespace gd
