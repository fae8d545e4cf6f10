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
//   TABLE : table.store(k, pniels) / table.load(k) for the per-lane window table, table.put_step(pniels) /
//           table.step() for the one value the build re-reads (k is public there);
//           table.lookup(idx) is the read of a digit's entry (public digits only: scalars that may be secret
//           are multiplied without a table, montgomery.hpp)
#pragma once
#include "point.hpp"
#include "sc14.hpp"

namespace gd {

// W-bit window of s' whose least significant bit is bit `pos` (pos + W <= 450).
template <int W, class BITS>
GD_FN uint32_t window_w(const BITS &bits, int pos) {
    const int k = pos >> 5, sh = pos & 31;
    uint32_t lo = bits.word(k) >> sh;
    uint32_t hi = sh > 32 - W ? bits.word(k + 1) << (32 - sh) : 0u;  // the window straddles two words
    return (lo | hi) & ((1u << W) - 1);
}
template <class BITS>
GD_FN uint32_t window5(const BITS &bits, int pos) { return window_w<5>(bits, pos); }

// digit w -> (table index, negate): index (w^inv)&(2^(W-1)-1) with inv = (w>>(W-1))-1; the entry
// holds (2*idx+1)*B; negate iff w < 2^(W-1)  (src/goldilocks.c:437-442 with W = 5).
template <int W>
GD_FN void signed_digit_w(uint32_t w, uint32_t &idx, bool &neg) {
    neg = w < (1u << (W - 1));
    idx = (neg ? ~w : w) & ((1u << (W - 1)) - 1);
}
GD_FN void signed_digit(uint32_t w, uint32_t &idx, bool &neg) { signed_digit_w<5>(w, idx, neg); }

// The reference's window plan: signed 5-bit windows, 90 of them, 16 table entries (src/goldilocks.c:405-465).
template <int W>
struct window_plan {
    static_assert(W == 5, "the recoding constant exists for 5-bit windows");
    static constexpr int ENTRIES = 1 << (W - 1);
    static constexpr int WINDOWS = (446 + W - 1) / W;             // 90
    static constexpr int TOP = W * (WINDOWS - 1);                 // 445
};
template <int W>
GD_FN sc sc_recode_window(const sc &s) { return sc_recode_signed(s); }

// multiples[k] = (2k+1)*B, k < 2^(W-1), as projective niels (src/goldilocks.c:382-403).
// The step 2B does not stay in registers next to the accumulator (that spills): the policy parks it
// (table.put_step) and hands it back every iteration (table.step()) -- in the table's own memory, slot
// ENTRIES, or wherever a kernel has room that is not behind its stores (kernels.hpp LdsStepTable).
template <int W, class TABLE>
GD_FN void build_window_table_w(TABLE &table, const pt &b) {
    constexpr int E = window_plan<W>::ENTRIES;
    pt twice = b;
    pt_double(twice, true);
    table.put_step(pt_to_pniels(twice));
    table.store(0, pt_to_pniels(b));
    pt acc = b;
#pragma unroll 1
    for (int k = 1; k < E; k++) {
        pt_add_pniels(acc, table.step(), false, true);
        table.store(k, pt_to_pniels(acc));
    }
}
template <class TABLE>
GD_FN void build_window_table(TABLE &table, const pt &b) { build_window_table_w<5>(table, b); }

// out = s * B with the window table already built and s' readable through `bits`.
// table.lookup(idx) is the digit's entry: a direct read for public digits, a scan of the whole
// table for secret ones (the policy decides).
template <int W, class BITS, class TABLE>
GD_FN pt ladder_varbase_w(const BITS &bits, const TABLE &table) {
    uint32_t idx;
    bool neg;
    signed_digit_w<W>(window_w<W>(bits, window_plan<W>::TOP), idx, neg);
    pt acc = pniels_to_pt(table.lookup(idx), neg);
#pragma unroll 1
    for (int pos = window_plan<W>::TOP - W; pos >= 0; pos -= W) {
        signed_digit_w<W>(window_w<W>(bits, pos), idx, neg);
#pragma unroll 1
        for (int j = 0; j < W - 1; j++) pt_double(acc, false);
        // the entry's reads are issued before the window's last doubling (public digits only reach this ladder):
        // a field operation's worth of arithmetic for the memory to hide behind
        pniels e = table.lookup(idx);
        gd_keep_order();
        pt_double(acc, true);
        // T is only needed by a following addition, i.e. never after the last window's
        // add -- except that the caller wants a complete extended point at pos == 0.
        pt_add_pniels(acc, e, neg, pos == 0);
    }
    return acc;
}
template <class BITS, class TABLE>
GD_FN pt ladder_varbase(const BITS &bits, const TABLE &table) { return ladder_varbase_w<5>(bits, table); }

// (s1*B, s2*B) for one base: the window table is built once and walked twice ("next" row f4;
// the reference's point_dual_scalarmul, src/goldilocks.c:543-642, gets there with a bucket
// method -- same two group elements).
template <int W, class BITS, class TABLE>
GD_FN void ladder_dual_w(pt &out1, pt &out2, const BITS &bits1, const BITS &bits2, const TABLE &table) {
    out1 = ladder_varbase_w<W>(bits1, table);
    out2 = ladder_varbase_w<W>(bits2, table);
}
template <class BITS, class TABLE>
GD_FN void ladder_dual(pt &out1, pt &out2, const BITS &bits1, const BITS &bits2, const TABLE &table) {
    ladder_dual_w<5>(out1, out2, bits1, bits2, table);
}

// Comb: S rounds; round i adds, for each of the N combs j, the entry selected by the T bits
// i + S*(k + T*j), k < T, of s' = (s + 2^(N*T*S) - 1)/2 mod q  (src/goldilocks.c:846-873 with the
// reference's N = T = 5, S = 18).  Entry idx of comb j is sum_k (+-) 2^(S(k + T j)) B, tooth T-1 always +,
// tooth k < T-1 + iff bit k of idx.
// COMB: comb.load(j, idx) -> affine niels entry 2^(T-1)*j + idx (our sign convention); COMB::plan says which comb.
template <int T, int N, int S>
struct comb_plan {
    static constexpr int TEETH = T, COMBS = N, SPACING = S, PER_COMB = 1 << (T - 1), ENTRIES = N << (T - 1);
    static_assert(T * N * S == 450 || T * N * S == 448, "recoding constants exist for 450 and 448 bits");
    static GD_MFN sc recode(const sc &s) { return T * N * S == 450 ? sc_recode_signed(s) : sc_recode_signed8(s); }
};
using comb_ref = comb_plan<5, 5, 18>;   // the reference's: 80 entries, 17 doublings + 89 additions
// 4 combs of 7 teeth, spacing 16: 256 entries (48 KiB as affine niels), 15 doublings + 63 additions -- 26 % fewer
// multiplications; for the library's own base point where the table has to be index-independent (kernels.hpp)
using comb_big = comb_plan<7, 4, 16>;
// 4 combs of 8 teeth, spacing 14: 512 entries (96 KiB), 13 doublings + 55 additions -- for a verification key that signs
// hundreds of a batch's signatures (kernels_verify.hip): twice the table to build, 9 % less to walk
using comb_wide = comb_plan<8, 4, 14>;
// 5 combs of 9 teeth, spacing 10 (450 bits): 1 280 entries (240 KiB), 9 doublings + 49 additions -- for a key that signs a
// thousand of them: two and a half times the wide comb's table, another 10 % less to walk
using comb_xwide = comb_plan<9, 5, 10>;

template <class PLAN, class BITS>
GD_FN uint32_t comb_teeth_of(const BITS &bits, int i, int j) {
    uint32_t tab = 0;
#pragma unroll
    for (int k = 0; k < PLAN::TEETH; k++) {
        const int bit = i + PLAN::SPACING * (k + PLAN::TEETH * j);
        if (bit < 446) tab |= ((bits.word(bit >> 5) >> (bit & 31)) & 1u) << k;
    }
    return tab;
}
template <class BITS>
GD_FN uint32_t comb_teeth(const BITS &bits, int i, int j) { return comb_teeth_of<comb_ref>(bits, i, j); }

template <class BITS, class COMB>
GD_FN pt ladder_comb(const BITS &bits, const COMB &comb) {
    using PLAN = typename COMB::plan;
    constexpr int S = PLAN::SPACING, N = PLAN::COMBS, T = PLAN::TEETH;
    uint32_t idx;
    bool neg;
    signed_digit_w<T>(comb_teeth_of<PLAN>(bits, S - 1, 0), idx, neg);
    pt acc = niels_to_pt(comb.load(0, idx), neg);
#pragma unroll 1
    for (int i = S - 1; i >= 0; i--) {
        if (i != S - 1) pt_double(acc, true);
#pragma unroll 1
        for (int j = (i == S - 1 ? 1 : 0); j < N; j++) {
            signed_digit_w<T>(comb_teeth_of<PLAN>(bits, i, j), idx, neg);
            niels e = comb.load(j, idx);
            // T feeds the next addition; the last add of a round is followed by a
            // doubling (which ignores T) unless it is the very last one.
            pt_add_niels(acc, e, neg, !(j == N - 1 && i));
        }
    }
    return acc;
}

// ---- the same walk for PUBLIC digits and a comb that is READ BY THE DIGIT (a verification key's comb in global
// memory, kernels_verify.hip).  ladder_comb above fetches a digit's T bits with T dependent LDS reads and only then
// asks for the entry, N * S times per multiplication, each time waited for (round 5's counters: the key-comb kernel
// spent 19 % of its wave cycles in s_waitcnt against the ladder kernel's 9 %).  Here the N * S digits are transposed out
// of the recoded scalar ONCE (all bit positions are compile-time constants: two instructions per bit), kept as 16-bit
// (index | sign << 15) in the order the walk consumes them, and the entry of digit t + 1 is requested BEFORE the
// addition of digit t -- as ladder_bwt_onto does for the base point's table.
// DIG: dig.put(word k, two digits) / dig.get(t) -> digit t.
template <class PLAN>
struct comb_digits {
    static constexpr int COUNT = PLAN::COMBS * PLAN::SPACING, WORDS = (COUNT + 1) / 2;
    // digit t of the walk: round i = S - 1 - t / N, comb j = t % N  (src/goldilocks.c:846-862 for N = T = 5, S = 18)
    static GD_MFN uint32_t digit(const sc &r, int t) {
        constexpr int T = PLAN::TEETH, N = PLAN::COMBS, S = PLAN::SPACING;
        const int i = S - 1 - t / N, j = t % N;
        uint32_t tab = 0;
#pragma unroll
        for (int k = 0; k < T; k++) {
            const int bit = i + S * (k + T * j);
            if (bit < 446) tab |= ((r.w[bit >> 5] >> (bit & 31)) & 1u) << k;
        }
        uint32_t idx;
        bool neg;
        signed_digit_w<T>(tab, idx, neg);
        return idx | (neg ? 0x8000u : 0u);
    }
    template <class DIG>
    static GD_MFN void store(DIG &dig, const sc &r) {
#pragma unroll
        for (int k = 0; k < WORDS; k++)
            dig.put(k, digit(r, 2 * k) | (2 * k + 1 < COUNT ? digit(r, 2 * k + 1) << 16 : 0u));
    }
};
template <class DIG, class COMB>
GD_FN pt ladder_comb_digits(const DIG &dig, const COMB &comb) {
    using PLAN = typename COMB::plan;
    constexpr int N = PLAN::COMBS, COUNT = comb_digits<PLAN>::COUNT;
    uint32_t d = dig.get(0);
    pt acc = niels_to_pt(comb.load(0, d & 0x7fffu), (d & 0x8000u) != 0);
    d = dig.get(1);
    niels next = comb.load(1 % N, d & 0x7fffu);
    int j = 1 % N;
#pragma unroll 1
    for (int t = 1; t < COUNT; t++) {
        const niels e = next;
        const bool neg = (d & 0x8000u) != 0, first = j == 0, last = j == N - 1;
        j = last ? 0 : j + 1;
        if (t + 1 < COUNT) {
            d = dig.get(t + 1);
            next = comb.load(j, d & 0x7fffu);
        }
        gd_keep_order();        // (the scheduler would otherwise sink the reads to right before their first use)
        if (first) pt_double(acc, true);
        // T feeds the next addition; a round's last addition is followed by a doubling (which ignores T) unless it
        // is the very last one, whose caller wants a complete extended point
        pt_add_niels(acc, e, neg, !(last && t + 1 < COUNT));
    }
    return acc;
}

// ---- the 4 x 7 x 16 comb of an ARBITRARY point P (a caller's table re-combed, kernels_fixed.hip; a verification
// key that signed many of a batch's signatures, kernels_verify.hip).  Tooth m (m < 28) is 2^(16 m) * P; entry
// e = 64 j + idx is T_(6+7j) + sum_{k<6} (+-) T_(k+7j), + iff bit k of idx, as an affine niels in our form.
// TEETH: teeth.load(m) -> pniels.
template <class PLAN, class TEETH>
GD_FN pt comb_entry_projective(const TEETH &teeth, uint32_t e) {
    const uint32_t j = e / PLAN::PER_COMB, idx = e % PLAN::PER_COMB;
    pt p = pniels_to_pt(teeth.load(PLAN::TEETH - 1 + PLAN::TEETH * j), false);
#pragma unroll 1
    for (uint32_t k = 0; k + 1 < (uint32_t)PLAN::TEETH; k++)
        pt_add_pniels(p, teeth.load(k + PLAN::TEETH * j), ((idx >> k) & 1u) == 0, true);
    return p;
}
template <class TEETH>
GD_FN pt comb_big_entry_projective(const TEETH &teeth, uint32_t e) { return comb_entry_projective<comb_big>(teeth, e); }
GD_FN niels comb_big_normalise(const pt &p) {
    const fe zi = fe_invert(fe_weak(fe_add(p.z, p.z)));
    niels n;
    n.a = fe_mul(fe_weak(fe_sub<2>(p.y, p.x)), zi);
    n.b = fe_mul(fe_weak(fe_add(p.x, p.y)), zi);
    n.cn = fe_mul(fe_mulw(p.t, TWO_EFF_D), zi);
    return n;
}
template <class TEETH>
GD_FN niels comb_big_entry(const TEETH &teeth, uint32_t e) { return comb_big_normalise(comb_big_entry_projective(teeth, e)); }

// Fixed-base, no doublings: s*B = sum_i (+-) T_i[idx_i] over the signed w-bit digits of the recoded scalar
// W = (s + 2^(w*windows) - 1)/2 mod q, with T_i[k] = (2k+1) * 2^(w*i) * B as affine niels, built once per device.
// The digit width w is the TABLE's (its header says so; goldilocks_amd_set_base_table_bits): 8 bits are 56 x 128
// entries (1.3 MiB) and 55 mixed additions; 16 bits 28 x 32 768 entries (168 MiB, Infinity-Cache resident) and 27;
// 24 bits 19 x 2^23 entries (28.5 GiB of the 288 GiB of HBM) and 18 -- instead of the comb's 17 doublings + 89
// additions.  The reference has no such table; results are the same group element (parity is on encodings).
// BWT: geom() -> BwtGeom, adjust() -> the recoding offset (2^(w*windows) - 1) mod q, load(geom, i, idx) -> niels.
struct BwtGeom {
    uint32_t bits, windows;
};
constexpr uint32_t BWT_BITS_MIN = 8, BWT_BITS_MAX = 24;
#ifdef __HIPCC__
#define GD_HD __host__ __device__ inline
#else
#define GD_HD inline
#endif
GD_HD constexpr uint32_t bwt_windows(uint32_t bits) { return (446 + bits - 1) / bits; }
GD_HD constexpr bool bwt_bits_supported(uint32_t bits) { return bits >= BWT_BITS_MIN && bits <= BWT_BITS_MAX && bits % 2 == 0; }
GD_HD constexpr uint64_t bwt_entries(uint32_t bits) { return (uint64_t)bwt_windows(bits) << (bits - 1); }
// the recoding offsets that exist: 8/14/16 bits span 448 bits, 10/18 span 450, 12/24 span 456, 20 span 460, 22 span 462
GD_FN sc bwt_adjust_for(uint32_t bits) {
    const uint32_t span = bits * bwt_windows(bits);
    return span == 450   ? sc_const(SC_ADJ)
           : span == 456 ? sc_const(SC_ADJ12)
           : span == 460 ? sc_const(SC_ADJ20)
           : span == 462 ? sc_const(SC_ADJ22)
                         : sc_const(SC_ADJ8);
}
template <class BWT>
GD_FN sc sc_recode_bwt(const sc &s, const BWT &bwt) {
    return sc_halve(sc_add(s, bwt.adjust()));
}
// (BITS holds a zero fifteenth word: the top digit of a span beyond 448 bits reaches into it, and a digit's second
// word is read whether it straddles or not)
template <class BITS>
GD_FN uint32_t window_bwt(const BITS &bits, uint32_t i, uint32_t w) {
    const uint32_t pos = w * i, k = pos >> 5, sh = pos & 31;
    const uint64_t two = (uint64_t)bits.word((int)k) | (uint64_t)bits.word((int)k + 1) << 32;
    return (uint32_t)(two >> sh) & ((1u << w) - 1);
}
GD_FN void signed_digit_bwt(uint32_t d, uint32_t w, uint32_t &idx, bool &neg) {
    const uint32_t per_window = 1u << (w - 1);
    neg = d < per_window;
    idx = (neg ? ~d : d) & (per_window - 1);
}
// acc += s*B through the same table: one mixed addition per digit onto a caller's accumulator (acc.t valid).
// The entry of the NEXT digit is requested before the current addition: a gather from a table of hundreds of MiB is
// a miss almost every time, and 28 of them in a row, each waited for, cost as much as 8 of the 28 additions
// (tools/verifyphases: 12.6 clocks per multiply-accumulate against the ladder's 9.7).
template <class BITS, class BWT>
GD_FN void ladder_bwt_onto(pt &acc, const BITS &bits, const BWT &bwt) {
    const BwtGeom g = bwt.geom();
    uint32_t idx;
    bool neg;
    signed_digit_bwt(window_bwt(bits, g.windows - 1, g.bits), g.bits, idx, neg);
    niels next = bwt.load(g, g.windows - 1, idx);
#pragma unroll 1
    for (uint32_t i = g.windows; i-- > 0;) {
        const niels e = next;
        const bool neg_e = neg;
        if (i > 0) {
            signed_digit_bwt(window_bwt(bits, i - 1, g.bits), g.bits, idx, neg);
            next = bwt.load(g, i - 1, idx);
        }
        pt_add_niels(acc, e, neg_e, true);
    }
}
// ladder_bwt with the next digit's entry requested an addition ahead (as ladder_bwt_onto): for a kernel with registers to
// spare -- k_verify_base_part: 230 of 256, nothing spilled, where ladder_bwt itself spilled 15 and waited for each of its
// 23 gathers (config 4: 7.14 against 7.18 - 7.23 ms, profiles/r06/ab_split_walk_and_base_prefetch.txt); at the register
// limit the 48 registers of the entry in flight are spilled instead (k_base_scalarmul + 77, k_x448 + 44, k_ed448_sign + 39).
template <class BITS, class BWT>
GD_FN pt ladder_bwt_ahead(const BITS &bits, const BWT &bwt) {
    const BwtGeom g = bwt.geom();
    uint32_t idx;
    bool neg;
    signed_digit_bwt(window_bwt(bits, g.windows - 1, g.bits), g.bits, idx, neg);
    pt acc = niels_to_pt(bwt.load(g, g.windows - 1, idx), neg);
    signed_digit_bwt(window_bwt(bits, g.windows - 2, g.bits), g.bits, idx, neg);
    niels next = bwt.load(g, g.windows - 2, idx);
#pragma unroll 1
    for (uint32_t i = g.windows - 1; i-- > 0;) {
        const niels e = next;
        const bool neg_e = neg;
        if (i > 0) {
            signed_digit_bwt(window_bwt(bits, i - 1, g.bits), g.bits, idx, neg);
            next = bwt.load(g, i - 1, idx);
        }
        pt_add_niels(acc, e, neg_e, true);
    }
    return acc;
}
template <class BITS, class BWT>
GD_FN pt ladder_bwt(const BITS &bits, const BWT &bwt) {
    const BwtGeom g = bwt.geom();
    uint32_t idx;
    bool neg;
    signed_digit_bwt(window_bwt(bits, g.windows - 1, g.bits), g.bits, idx, neg);
    pt acc = niels_to_pt(bwt.load(g, g.windows - 1, idx), neg);
#pragma unroll 1
    for (uint32_t i = g.windows - 1; i-- > 0;) {
        signed_digit_bwt(window_bwt(bits, i, g.bits), g.bits, idx, neg);
        pt_add_niels(acc, bwt.load(g, i, idx), neg, true);
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
        auto bits = mk(COMB::plan::recode(s), 0);
        return ladder_comb(bits, comb);
    }
    template <class MK>
    GD_MFN pt mul_ahead(const sc &s, MK &mk) const { return mul(s, mk); }
    template <class MK>
    GD_MFN void add_to(pt &acc, const sc &s, MK &mk) const { acc = pt_add(acc, mul(s, mk), false); }
};
template <class BWT>
struct FixedBwt {
    const BWT &bwt;
    template <class MK>
    GD_MFN pt mul(const sc &s, MK &mk) const {
        auto bits = mk(sc_recode_bwt(s, bwt), 0);
        return ladder_bwt(bits, bwt);
    }
    template <class MK>
    GD_MFN pt mul_ahead(const sc &s, MK &mk) const {      // (public scalars, registers to spare: ladder_bwt_ahead)
        auto bits = mk(sc_recode_bwt(s, bwt), 0);
        return ladder_bwt_ahead(bits, bwt);
    }
    template <class MK>
    GD_MFN void add_to(pt &acc, const sc &s, MK &mk) const {
        auto bits = mk(sc_recode_bwt(s, bwt), 0);
        ladder_bwt_onto(acc, bits, bwt);
    }
};

// out = s1*P1 + s2*P2, both through window tables of width W (src/goldilocks.c:467-541 with W = 5).
template <int W, class BITS, class TABLE1, class TABLE2>
GD_FN pt ladder_double_w(const BITS &bits1, const TABLE1 &t1, const BITS &bits2, const TABLE2 &t2) {
    uint32_t idx;
    bool neg;
    signed_digit_w<W>(window_w<W>(bits1, window_plan<W>::TOP), idx, neg);
    pt acc = pniels_to_pt(t1.lookup(idx), neg);
    signed_digit_w<W>(window_w<W>(bits2, window_plan<W>::TOP), idx, neg);
    pt_add_pniels(acc, t2.lookup(idx), neg, false);
#pragma unroll 1
    for (int pos = window_plan<W>::TOP - W; pos >= 0; pos -= W) {
#pragma unroll 1
        for (int j = 0; j < W; j++) pt_double(acc, j == W - 1);
        if constexpr (TABLE1::direct && TABLE2::direct) {   // both entries' loads are issued before the first addition
            uint32_t idx2;
            bool neg2;
            signed_digit_w<W>(window_w<W>(bits1, pos), idx, neg);
            signed_digit_w<W>(window_w<W>(bits2, pos), idx2, neg2);
            const pniels e1 = t1.lookup(idx);
            const pniels e2 = t2.lookup(idx2);
            pt_add_pniels(acc, e1, neg, true);
            pt_add_pniels(acc, e2, neg2, pos == 0);
        } else {
            signed_digit_w<W>(window_w<W>(bits1, pos), idx, neg);
            pt_add_pniels(acc, t1.lookup(idx), neg, true);
            signed_digit_w<W>(window_w<W>(bits2, pos), idx, neg);
            pt_add_pniels(acc, t2.lookup(idx), neg, pos == 0);
        }
    }
    return acc;
}
template <class BITS, class TABLE1, class TABLE2>
GD_FN pt ladder_double(const BITS &bits1, const TABLE1 &t1, const BITS &bits2, const TABLE2 &t2) {
    return ladder_double_w<5>(bits1, t1, bits2, t2);
}
// The same with `nw` 5-bit windows instead of 90 (nw must be the same in every lane of a wave): for the
// half-size scalars of verification (lattice.hpp).  bits: words of (s + 2^(5 nw) - 1) / 2 for odd INTEGER s.
// Public data (TABLE::direct): the first entry of a window is requested BEFORE the window's last doubling and
// the second one before the first addition, so each read has a field operation's worth of arithmetic to hide
// behind (with both requested after the doublings the first addition waited for memory 44 times per signature).
// flip1: table 1 holds the multiples of P where those of -P are meant (a table shared by the signatures of one key
// serves both signs of tau): every digit of scalar 1 changes sign.
template <class BITS, class TABLE1, class TABLE2>
GD_FN pt ladder_double_var(const BITS &bits1, const TABLE1 &t1, bool flip1, const BITS &bits2, const TABLE2 &t2, int nw) {
    uint32_t idx;
    bool neg;
    signed_digit(window5(bits1, 5 * (nw - 1)), idx, neg);
    pt acc = pniels_to_pt(t1.lookup(idx), neg != flip1);
    signed_digit(window5(bits2, 5 * (nw - 1)), idx, neg);
    pt_add_pniels(acc, t2.lookup(idx), neg, false);
#pragma unroll 1
    for (int pos = 5 * (nw - 2); pos >= 0; pos -= 5) {
#pragma unroll 1
        for (int j = 0; j < 4; j++) pt_double(acc, false);
        uint32_t idx2;
        bool neg2;
        signed_digit(window5(bits1, pos), idx, neg);
        signed_digit(window5(bits2, pos), idx2, neg2);
        const pniels e1 = t1.lookup(idx);
        gd_keep_order();        // (the scheduler would otherwise sink the reads to right before their first use)
        pt_double(acc, true);
        const pniels e2 = t2.lookup(idx2);
        gd_keep_order();
        pt_add_pniels(acc, e1, neg != flip1, true);
        pt_add_pniels(acc, e2, neg2, true);
    }
    return acc;
}

}  // namespace gd
