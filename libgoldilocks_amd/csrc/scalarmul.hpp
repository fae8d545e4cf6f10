// scalarmul.hpp -- the scalar-multiplication ladders, one operation per lane.
//
// The loop structure restates the reference's three scalarmuls:
//   variable base, signed 5-bit fixed windows   src/goldilocks.c:405-465
//   fixed base, 5x5x18 signed comb              src/goldilocks.c:830-877
//   double base a*P + b*Q, interleaved windows  src/goldilocks.c:467-541
// All three walk the recoded scalar s' = (s + 2^450 - 1)/2 mod q, whose 5-bit digit
// w in [0,32) stands for the odd signed digit 2w - 31.
//
// The ladders are templates over two small policies so that the same code is
// (a) instantiated by the HIP kernels with HBM/LDS-backed storage and (b) run on
// the host by tests/hostsim with plain arrays (checker only, never shipped):
//   BITS  : bits.word(k) -> k-th 32-bit word of s' (k < 15, word 14 is zero)
//   TABLE : table.store(k, pniels) / table.load(k) for the per-lane window table (17 slots:
//           16 entries + one scratch slot used while the table is built)
#pragma once
#include "point.hpp"
#include "sc14.hpp"

namespace gd {

// 5-bit window of s' whose least significant bit is bit `pos` (pos <= 445).
template <class BITS>
GD_FN uint32_t window5(const BITS &bits, int pos) {
    const int k = pos >> 5, sh = pos & 31;
    uint32_t lo = bits.word(k) >> sh;
    uint32_t hi = sh > 27 ? bits.word(k + 1) << (32 - sh) : 0u;  // sh > 27: window straddles
    return (lo | hi) & 31u;
}

// digit w -> (table index, negate): index (w^inv)&15 with inv = (w>>4)-1; entry
// holds (2*idx+1)*B; negate iff w < 16  (src/goldilocks.c:437-442).
GD_FN void signed_digit(uint32_t w, uint32_t &idx, bool &neg) {
    neg = w < 16;
    idx = (neg ? ~w : w) & 15u;
}

// multiples[k] = (2k+1)*B, k < 16, as projective niels (src/goldilocks.c:382-403).
// The step 2B is parked in slot 16 of the lane's table memory and re-read every iteration, as the
// ladder re-reads its entries: holding it in registers next to the accumulator spills.
template <class TABLE>
GD_FN void build_window_table(TABLE &table, const pt &b) {
    pt twice = b;
    pt_double(twice, true);
    table.store(16, pt_to_pniels(twice));
    table.store(0, pt_to_pniels(b));
    pt acc = b;
#pragma unroll 1
    for (int k = 1; k < 16; k++) {
        pt_add_pniels(acc, table.load(16), false, true);
        table.store(k, pt_to_pniels(acc));
    }
}

// out = s * B with the window table already built and s' readable through `bits`.
template <class BITS, class TABLE>
GD_FN pt ladder_varbase(const BITS &bits, const TABLE &table) {
    uint32_t idx;
    bool neg;
    signed_digit(window5(bits, 445), idx, neg);
    pt acc = pniels_to_pt(table.load(idx), neg);
#pragma unroll 1
    for (int pos = 440; pos >= 0; pos -= 5) {
        signed_digit(window5(bits, pos), idx, neg);
#pragma unroll 1
        for (int j = 0; j < 5; j++) pt_double(acc, j == 4);
        pniels e = table.load(idx);
        // T is only needed by a following addition, i.e. never after the last window's
        // add -- except that the caller wants a complete extended point at pos == 0.
        pt_add_pniels(acc, e, neg, pos == 0);
    }
    return acc;
}

// (s1*B, s2*B) for one base: the window table is built once and walked twice ("next" row f4;
// the reference's point_dual_scalarmul, src/goldilocks.c:543-642, gets there with a bucket
// method -- same two group elements).
template <class BITS, class TABLE>
GD_FN void ladder_dual(pt &out1, pt &out2, const BITS &bits1, const BITS &bits2, const TABLE &table) {
    out1 = ladder_varbase(bits1, table);
    out2 = ladder_varbase(bits2, table);
}

// Comb: 18 rounds; round i adds, for each of the 5 combs j, the entry selected by
// bits i + 18*(k + 5j), k < 5, of s'  (src/goldilocks.c:846-873).
// COMB: comb.load(j, idx) -> affine niels entry 16*j + idx (our sign convention).
template <class BITS>
GD_FN uint32_t comb_teeth(const BITS &bits, int i, int j) {
    uint32_t tab = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int bit = i + 18 * (k + 5 * j);
        if (bit < 446) tab |= ((bits.word(bit >> 5) >> (bit & 31)) & 1u) << k;
    }
    return tab;
}

template <class BITS, class COMB>
GD_FN pt ladder_comb(const BITS &bits, const COMB &comb) {
    uint32_t idx;
    bool neg;
    signed_digit(comb_teeth(bits, 17, 0), idx, neg);
    pt acc = niels_to_pt(comb.load(0, idx), neg);
#pragma unroll 1
    for (int i = 17; i >= 0; i--) {
        if (i != 17) pt_double(acc, true);
#pragma unroll 1
        for (int j = (i == 17 ? 1 : 0); j < 5; j++) {
            signed_digit(comb_teeth(bits, i, j), idx, neg);
            niels e = comb.load(j, idx);
            // T feeds the next addition; the last add of a round is followed by a
            // doubling (which ignores T) unless it is the very last one.
            pt_add_niels(acc, e, neg, !(j == 4 && i));
        }
    }
    return acc;
}

// Fixed-base, no doublings: s*B = sum_i (+-) T_i[idx_i] over the signed BWT_BITS-bit digits of the
// recoded scalar W = (s + 2^(BWT_BITS*BWT_WINDOWS) - 1)/2 mod q, with T_i[k] = (2k+1) * 2^(BWT_BITS*i) * B
// as affine niels, built once per device.  With 8-bit digits: 56 x 128 entries (1.3 MiB), 55 mixed
// additions; with 16-bit digits (the default): 28 x 32768 entries (168 MiB, Infinity-Cache resident),
// 27 mixed additions -- instead of the comb's 17 doublings + 89 additions.  The reference
// has no such table; results are the same group element (parity is on encodings).
// BWT: bwt.load(i, idx) -> niels.
#ifndef GD_BWT_BITS
#define GD_BWT_BITS 16
#endif
constexpr int BWT_BITS = GD_BWT_BITS;
static_assert(BWT_BITS == 8 || BWT_BITS == 10 || BWT_BITS == 12 || BWT_BITS == 14 || BWT_BITS == 16,
              "recoding constants exist for 8/14/16-bit (2^448-1), 10-bit (2^450-1) and 12-bit (2^456-1) digits");
constexpr int BWT_WINDOWS = (446 + BWT_BITS - 1) / BWT_BITS;   // 56 or 45
constexpr int BWT_PER_WINDOW = 1 << (BWT_BITS - 1);             // entries per window: 128 or 512
GD_FN sc sc_recode_bwt(const sc &s) {
    return BWT_BITS == 10 ? sc_recode_signed(s) : BWT_BITS == 12 ? sc_recode_signed12(s) : sc_recode_signed8(s);
}
template <class BITS>
GD_FN uint32_t window_bwt(const BITS &bits, int i) {
    const int pos = BWT_BITS * i, k = pos >> 5, sh = pos & 31;
    uint32_t lo = bits.word(k) >> sh;
    uint32_t hi = sh + BWT_BITS > 32 ? bits.word(k + 1) << (32 - sh) : 0u;   // the digit straddles two words
    return (lo | hi) & ((1u << BWT_BITS) - 1);
}
GD_FN void signed_digit_bwt(uint32_t w, uint32_t &idx, bool &neg) {
    neg = w < (uint32_t)BWT_PER_WINDOW;
    idx = (neg ? ~w : w) & (uint32_t)(BWT_PER_WINDOW - 1);
}
template <class BITS, class BWT>
GD_FN pt ladder_bwt(const BITS &bits, const BWT &bwt) {
    uint32_t idx;
    bool neg;
    signed_digit_bwt(window_bwt(bits, BWT_WINDOWS - 1), idx, neg);
    pt acc = niels_to_pt(bwt.load(BWT_WINDOWS - 1, idx), neg);
#pragma unroll 1
    for (int i = BWT_WINDOWS - 2; i >= 0; i--) {
        signed_digit_bwt(window_bwt(bits, i), idx, neg);
        pt_add_niels(acc, bwt.load(i, idx), neg, true);
    }
    return acc;
}

// Fixed-base multiplication behind one interface, so that sign / derive / verify do not care which
// table serves the base point.  MK: mk(recoded scalar, slot) -> BITS (LDS-backed on the device).
template <class COMB>
struct FixedComb {
    const COMB &comb;
    template <class MK>
    GD_MFN pt mul(const sc &s, MK &mk) const {
        auto bits = mk(sc_recode_signed(s), 0);
        return ladder_comb(bits, comb);
    }
};
template <class BWT>
struct FixedBwt {
    const BWT &bwt;
    template <class MK>
    GD_MFN pt mul(const sc &s, MK &mk) const {
        auto bits = mk(sc_recode_bwt(s), 0);
        return ladder_bwt(bits, bwt);
    }
};

// out = s1*P1 + s2*P2, both through 16-entry window tables (src/goldilocks.c:467-541).
template <class BITS, class TABLE1, class TABLE2>
GD_FN pt ladder_double(const BITS &bits1, const TABLE1 &t1, const BITS &bits2, const TABLE2 &t2) {
    uint32_t idx;
    bool neg;
    signed_digit(window5(bits1, 445), idx, neg);
    pt acc = pniels_to_pt(t1.load(idx), neg);
    signed_digit(window5(bits2, 445), idx, neg);
    pt_add_pniels(acc, t2.load(idx), neg, false);
#pragma unroll 1
    for (int pos = 440; pos >= 0; pos -= 5) {
#pragma unroll 1
        for (int j = 0; j < 5; j++) pt_double(acc, j == 4);
        signed_digit(window5(bits1, pos), idx, neg);
        pt_add_pniels(acc, t1.load(idx), neg, true);
        signed_digit(window5(bits2, pos), idx, neg);
        pt_add_pniels(acc, t2.load(idx), neg, pos == 0);
    }
    return acc;
}

}  // namespace gd
