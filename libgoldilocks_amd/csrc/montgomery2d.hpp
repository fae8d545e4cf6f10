// montgomery2d.hpp -- s1*P1 + s2*P2 for SECRET scalars on ONE chain: a two-dimensional differential ladder.
//
// The reference's goldilocks_448_point_double_scalarmul (src/goldilocks.c:467-541) walks both signed-window tables on
// one doubling chain -- 1.32 x a single multiplication -- and reads them with constant_time_lookup.  A per-lane table
// cannot be read index-independently on this machine (16 entries of 256 bytes per digit and lane), which is why the
// library multiplies a variable base by a secret scalar with a table-free Montgomery ladder (montgomery.hpp); two
// scalars were two such ladders until round 6: 2.0 x.  This is the table-free counterpart of the shared doubling chain,
// the uniform two-dimensional binary chain of D. J. Bernstein's "Differential addition chains" (2006, section 4) in the
// form it takes when the four corners of a unit square are the candidates:
//
//   with (A, B) the scalars' leading bits so far, the chain holds THREE of the four corners (A+x) P1 + (B+y) P2,
//   x, y in {0, 1}: the one with both coefficients even ("ee"), the one with both odd ("oo") and ONE of the two mixed
//   ones.  Appending a bit pair d = (alpha, beta) replaces them by
//       ee' = 2 * corner[d]                     (a doubling)
//       oo' = ee + oo                           (difference +-(P1 + P2) or +-(P1 - P2), by the parity class of the
//                                                bit pair before)
//       mixed' = mixed + (ee or oo)             (difference +-P1 or +-P2)
//   -- one doubling and two differential additions whatever the bits: 8 M + 6 S + 1 small multiplication per bit
//   against the 10 M + 8 S + 2 of two ladders.  Which mixed corner is kept is not free: the corner that is dropped must
//   never be the one the next doubling needs, which fixes it from the LOWER bits: the dropped corner at a level is the
//   complement of the most recent lower bit pair of the other parity class (derivation and an integer model of the
//   whole chain: docs/history/r06.md; tests/hostsim runs the chain on the host, against the CPU restatement).  That control stream
//   c -- one bit per level -- is computed from the bottom in a pre-pass (ml2_control); like the scalars' own bits it
//   only ever feeds selections.
//
// Everything is x-only on the Montgomery model u = (Y + Z) / (Y - Z), as in montgomery.hpp; the four differences
// u(P1), u(P2), u(P1 + P2), u(P1 - P2) are affine (one shared inversion for the four) and live in LDS; the result is
// recovered from u(R) and u(R + P1) by the same Okeya-Sakurai step (ml_recover).  The differential addition is valid
// for every pair of operands as long as the DIFFERENCE is neither the identity nor the 2-torsion point, so the
// exceptional inputs are dealt with BEFORE the chain, by substitution (ml2_effective): a trivial P1 or P2, or
// P1 = +-P2 up to 2-torsion, turn into a chain on two independent points with one scalar zero.
#pragma once
#include "montgomery.hpp"

namespace gd {

// ---- exceptional inputs by substitution.  G: the curve's base point (any point of full order would do).
struct Ml2Inputs {
    pt p1, p2;      // neither trivial, p1 != +-p2 up to 2-torsion
    sc a, b;        // a*p1 + b*p2 == s1*P1 + s2*P2 up to goldilocks_448_point_eq
};
GD_FN bool ml2_same_up_to_sign(const pt &p, const pt &q) {      // p == +-q up to 2-torsion (src/goldilocks.c:644-653, both signs)
    const fe l = fe_mul(p.y, q.x), r = fe_mul(q.y, p.x);
    return fe_eq(l, r) | fe_is_zero(fe_weak(fe_add(l, r)));
}
GD_FN pt ml2_pick(const pt &a, const pt &b, bool pick_b) {
    pt r;
    r.x = fe_select(a.x, b.x, pick_b);
    r.y = fe_select(a.y, b.y, pick_b);
    r.z = fe_select(a.z, b.z, pick_b);
    r.t = fe_select(a.t, b.t, pick_b);
    return r;
}
GD_FN sc ml2_pick(const sc &a, const sc &b, bool pick_b) {
    sc r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.w[i] = pick_b ? b.w[i] : a.w[i];
    return r;
}
GD_FN Ml2Inputs ml2_effective(const pt &P1, const pt &P2, const sc &s1, const sc &s2, const pt &G) {
    pt G2 = G;
    pt_double(G2, true);
    const bool t1 = fe_is_zero(P1.x), t2 = fe_is_zero(P2.x);                  // the identity or (0, -1): contributes nothing
    const fe l = fe_mul(P1.y, P2.x), r = fe_mul(P2.y, P1.x);
    const bool both = !t1 & !t2;
    const bool same = both & fe_eq(l, r), opposite = both & !same & fe_is_zero(fe_weak(fe_add(l, r)));
    const bool drop2 = t2 | same | opposite;                                  // P2's share moves into a (or is nothing)
    Ml2Inputs in;
    in.a = ml2_pick(ml2_pick(ml2_pick(s1, sc_add(s1, s2), same), sc_sub(s1, s2), opposite), sc_zero(), t1);
    in.b = ml2_pick(s2, sc_zero(), drop2);
    // the stand-ins: G and 2 G are independent of each other; a point is +-G or +-2G, not both (G has prime order q > 3)
    const bool p2_is_g = ml2_same_up_to_sign(P2, G), p1_is_g = ml2_same_up_to_sign(P1, G);
    const pt stand1 = ml2_pick(G, G2, !t2 & p2_is_g);                         // replaces a trivial P1 beside a live P2
    const pt stand2 = ml2_pick(ml2_pick(G, G2, !t1 & p1_is_g), G2, t1);       // replaces a dropped P2 (2 G beside the stand-in G)
    in.p1 = ml2_pick(P1, stand1, t1);
    in.p2 = ml2_pick(P2, stand2, drop2);
    return in;
}
// (when P1 is trivial AND P2 is live and equal to +-G, P1's stand-in is 2 G; when both are trivial the pair is G, 2 G)

// ---- the control stream: bit i of c says which mixed corner is DROPPED after level i -- 1: the one that differs from the
// level's bit pair in x, (~alpha_i, beta_i); 0: (alpha_i, ~beta_i).  Levels 0 .. 446 (the scalars have 446 bits: level
// 446 is the pair (0, 0) above them, which fixes the chain's first mixed corner).  From the bottom:
//   c_0 = beta_0 (so that R and R + P1 are both there at the end), and with dA, dB = whether alpha, beta change from
//   level i to level i + 1:   c_{i+1} = dA == dB ? c_i ^ dA : dB.
GD_FN void ml2_control(uint32_t c[14], const sc &a, const sc &b) {
    uint32_t cur = b.w[0] & 1u;                     // c_i as we go up
    uint32_t pa = a.w[0] & 1u, pb = cur;
    // (the words by compile-time index: a rolled loop over them would put the scalars into scratch memory, addressed by
    // a register)
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint32_t out = 0;
        const uint32_t wa = a.w[k], wb = b.w[k], na = k < 13 ? a.w[k + 1] : 0u, nb = k < 13 ? b.w[k + 1] : 0u;
#pragma unroll 1
        for (int j = 0; j < 32; j++) {
            out |= cur << j;
            // bit i + 1 (the next word's bit 0 at j == 31; zero beyond the scalars' 448 bits)
            const uint32_t xa = j < 31 ? (wa >> (j + 1)) & 1u : na & 1u, xb = j < 31 ? (wb >> (j + 1)) & 1u : nb & 1u;
            const uint32_t dA = xa ^ pa, dB = xb ^ pb;
            cur = dA == dB ? cur ^ dA : dB;
            pa = xa;
            pb = xb;
        }
        c[k] = out;
    }
}

// ---- the chain.  State: three points (x : z), products (mag 1, pairable) as in MlStateS.
struct Ml2State {
    sfp xe, ze, xo, zo, xm, zm;
};
// DIFFS: diffs.pair(k, pick_second) -> one of the two affine differences of pair k as an sfp, read index-independently
// (both are read, one is kept): pair 0 = (u(P1), u(P2)), pair 1 = (u(P1 + P2), u(P1 - P2)).
template <class DIFFS>
GD_FN void ml2_step(Ml2State &st, const DIFFS &diffs, bool t_oo, bool t_mixed, bool x_oo, bool use_p2, bool use_minus) {
    const sfp s_e = sfe_add(st.xe, st.ze), s_o = sfe_add(st.xo, st.zo), s_m = sfe_add(st.xm, st.zm);   // mag 2, pairable
    const sfs d_e = sfe_sub(st.xe, st.ze), d_o = sfe_sub(st.xo, st.zo), d_m = sfe_sub(st.xm, st.zm);   // mag 1, signed
    // the corner that is doubled, and the even-or-odd corner the mixed one is added to
    const sfp s_t = sfe_select(sfe_select(s_e, s_o, t_oo), s_m, t_mixed);
    const sfs d_t = sfe_select(sfe_select(d_e, d_o, t_oo), d_m, t_mixed);
    const sfp s_x = sfe_select(s_e, s_o, x_oo);
    const sfs d_x = sfe_select(d_e, d_o, x_oo);
    {   // oo' = ee + oo
        const sfp da = sfe_mul(s_e, d_o), cb = sfe_mul(s_o, d_e);                                       // 2 x 1
        st.zo = sfe_mul(sfe_sqr<false>(sfe_sub(da, cb)), s_multiplier(diffs.pair(1, use_minus)));
        st.xo = sfe_sqr<true>(sfe_add(da, cb));
    }
    {   // mixed' = mixed + (ee | oo)
        const sfp da = sfe_mul(s_m, d_x), cb = sfe_mul(s_x, d_m);
        st.zm = sfe_mul(sfe_sqr<false>(sfe_sub(da, cb)), s_multiplier(diffs.pair(0, use_p2)));
        st.xm = sfe_sqr<true>(sfe_add(da, cb));
    }
    {   // ee' = 2 * (ee | oo | mixed)
        const sfp aa = sfe_sqr<true>(s_t), bb = sfe_sqr<false>(d_t);
        const sfp caa = sfe_mulw(aa, (int32_t)ML_C);
        const sfs e = sfe_sub(aa, bb);
        st.xe = sfe_mul(caa, bb);
        st.ze = sfe_mul(sfe_add(caa, e), e);
    }
}

// bits_a, bits_b, bits_c: .word(k) of the (reduced) scalars and of the control stream.  p1: the first point (for the
// recovery); u1: u(p1); the four differences behind `diffs`.
// P1 is asked for AFTER the chain (p1_again()): a kernel derives it again rather than keep its 64 registers alive across
// the 446 steps.
template <class BITS, class DIFFS, class P1SRC>
GD_FN pt ml2_double_scalarmul_core(const fe &u1, const BITS &bits_a, const BITS &bits_b, const BITS &bits_c, const DIFFS &diffs,
                                   P1SRC &&p1_again) {
    Ml2State st;
    st.xe = sfe_from_fe(fe_one());
    st.ze = sfe_from_fe(fe_zero());
    st.xo = diffs.pair(1, false);                   // oo = P1 + P2
    st.zo = sfe_from_fe(fe_one());
    // level 446's control bit: 1 drops (1, 0) = P1, the chain starts with P2 as its mixed corner
    st.xm = diffs.pair(0, ((bits_c.word(13) >> (ML_BITS & 31)) & 1u) != 0);
    st.zm = sfe_from_fe(fe_one());
    bool prev_a = false, prev_b = false;            // the bit pair of the level above
    uint32_t last_a = 0, last_b = 0;
#pragma unroll 1
    for (int wi = (ML_BITS - 1) >> 5; wi >= 0; wi--) {
        const int top = wi == ((ML_BITS - 1) >> 5) ? ((ML_BITS - 1) & 31) : 31;
        uint32_t wa = bits_a.word(wi) << (31 - top), wb = bits_b.word(wi) << (31 - top), wc = bits_c.word(wi) << (31 - top);
#pragma unroll 1
        for (int j = top; j >= 0; j--) {
            const bool al = (int32_t)wa < 0, be = (int32_t)wb < 0, use_p2 = (int32_t)wc < 0;
            wa <<= 1;
            wb <<= 1;
            wc <<= 1;
            const bool eq_a = al == prev_a, eq_b = be == prev_b;
            const bool x_ee = use_p2 ? eq_a : eq_b;
            ml2_step(st, diffs, !eq_a & !eq_b, eq_a != eq_b, !x_ee, use_p2, prev_a != prev_b);
            prev_a = al;
            prev_b = be;
        }
    }
    last_a = prev_a ? 1u : 0u;
    last_b = prev_b ? 1u : 0u;
    // R = the corner (0, 0), R + P1 = the corner (1, 0): by the last bit pair, R is ee / mixed / mixed / oo and R + P1 is
    // mixed / ee / oo / mixed for (0,0) / (1,0) / (0,1) / (1,1)
    const fe xe = sfe_to_fe(st.xe), ze = sfe_to_fe(st.ze), xo = sfe_to_fe(st.xo), zo = sfe_to_fe(st.zo), xm = sfe_to_fe(st.xm),
             zm = sfe_to_fe(st.zm);
    const bool same = last_a == last_b, b1 = last_b != 0;
    const fe X1 = fe_select(xm, fe_select(xe, xo, b1), same), Z1 = fe_select(zm, fe_select(ze, zo, b1), same);
    const fe X2 = fe_select(fe_select(xe, xo, b1), xm, same), Z2 = fe_select(fe_select(ze, zo, b1), zm, same);
    return ml_recover(p1_again(), u1, X1, Z1, X2, Z2);
}
template <class BITS, class DIFFS>
GD_FN pt ml2_double_scalarmul(const pt &p1, const fe &u1, const BITS &bits_a, const BITS &bits_b, const BITS &bits_c, const DIFFS &diffs) {
    return ml2_double_scalarmul_core(u1, bits_a, bits_b, bits_c, diffs, [&]() { return p1; });
}

}  // namespace gd
