// montgomery.hpp -- variable-base multiplication WITHOUT a table: the index-independent default of
//   goldilocks_448_point_scalarmul                 src/goldilocks.c:405-465
// The reference reads its 16-entry window table through constant_time_lookup (src/include/constant_time.h:134-183:
// every entry for every digit).  One operation per lane, a lane's table lives in HBM and such a scan is what
// the kernel then waits for (200 GB per 2^20 operations).  This path needs no table at
// all: the Montgomery ladder (the shape of the reference's own goldilocks_x448, src/goldilocks.c:1006-1076)
// on the Montgomery model of the curve the reference computes on,
//     E: -x^2 + y^2 = 1 + d' x^2 y^2, d' = -39082      <->      M: B v^2 = u^3 + A u^2 + u,
//     u = (y + 1)/(y - 1),  v = -u/x,   A = 2(d' - 1)/(d' + 1) = 78166/39081,   (A - 2)/4 = 1/39081,
// followed by Okeya-Sakurai recovery of v and the map back to extended coordinates.  Every step is
// 5M + 4S + one multiplication by 39081 whatever the scalar; the only data-dependent instructions are selects.
//
// What the ladder computes is (s mod q) * P exactly.  The reference computes (s + m q) * P for an integer
// m that depends on its recoding (src/goldilocks.c:420-438); on the points its API produces -- the
// subgroup 2E, where q * P is the identity or the 2-torsion point (0, -1) -- the two agree up to that
// 2-torsion point, which is the equivalence goldilocks_448_point_eq and the encodings are defined on
// (src/goldilocks.c:644-653).
//
// Exceptional cases, all handled by selects on masks (no branch on the scalar):
//   * P in {identity, (0,-1)} (X = 0): u is infinite or zero; the result is the identity;
//   * s*P in {identity, (0,-1)}: the ladder's (X1 : Z1) has a zero; the result is the identity;
//   * (s+1)*P in {identity, (0,-1)} (s = q - 1): (X2 : Z2) has a zero and the recovery formula
//     degenerates; the result is -P.
#pragma once
#include "point.hpp"
#include "sc14.hpp"
#include "gf28s.hpp"
#include "tables_generated.h"

namespace gd {

constexpr uint32_t ML_C = 39081;            // 1 - d' ... the ladder's small constant: ((A - 2)/4)^-1
constexpr uint32_t ML_2AC = 156332;         // 2 A * 39081
constexpr int ML_BITS = 446;                // bits of q

// u(B) of the curve's base point (tools/gen_tables.py; tests/test_tables.py re-derives it): what
// goldilocks_448_direct_scalarmul multiplies when an encoding does not decode (src/goldilocks.c:898) -- a constant, so
// that fallback needs no inversion of its own inside a divergent branch
GD_CONST uint32_t ML_U_BASE28[16] = GD_POINT_BASE_U28_INIT;
GD_FN fe ml_u_base() {
    fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.v[i] = ML_U_BASE28[i];
    return r;
}

// What the ladder needs of the base point besides 1/(Y - Z): computed twice (once to learn the denominator
// that goes into the lane's shared inversion, once when the ladder runs) rather than parked in memory.
GD_FN fe ml_denominator(const pt &b) { return fe_weak(fe_sub<2>(b.y, b.z)); }   // Y - Z: zero iff P is the identity

// One ladder step on (x2 : z2) = k P, (x3 : z3) = (k+1) P with the conditional swap folded in, on the UNSIGNED limbs of
// gf28.hpp: the library's step until round 5, kept as the reference side of tools/stepbench's A/B (the kernels run
// ml_step_sel_s below).  Exchanging the two pairs exchanges DA and CB, so DA + CB and
// (DA - CB)^2 -- the new (x3 : z3) -- do not depend on the swap at all: only the pair that is DOUBLED has to be
// selected, i.e. one sum and one difference (32 selects per step instead of 64).  The differences are selected
// unreduced (mag 3: they multiply the other pair's sum, mag 2, as they are) and the selected one is reduced once.
GD_FN void ml_step_sel(fe &x2, fe &z2, fe &x3, fe &z3, const fe &x1, bool sw) {
    const fe s2 = fe_add(x2, z2), s3 = fe_add(x3, z3);              // mag 2
    const fe d2 = fe_sub<2>(x2, z2), d3 = fe_sub<2>(x3, z3);        // mag 3 (times mag 2 only)
    const fe da = fe_mul(s2, d3);                                   // DA (or CB, swapped: the same pair)
    const fe cb = fe_mul(s3, d2);
    const fe t1 = fe_select(s2, s3, sw);                            // A = sum of the pair to double      mag 2
    const fe t2 = fe_weak(fe_select(d2, d3, sw));                   // B = its difference                 mag 1
    const fe dm = fe_weak(fe_sub<2>(da, cb));                       // +-(DA - CB)                        mag 1
    z3 = fe_mul(fe_sqr(dm), x1);                                    // z3 = x1 (DA - CB)^2   (x1 second: its half sums are loop-invariant)
    x3 = fe_sqr(fe_add(da, cb));                                    // x3 = (DA + CB)^2       (input mag 2)
    const fe aa = fe_sqr(t1);                                       // AA                     (input mag 2)
    const fe bb = fe_sqr(t2);                                       // BB
    const fe caa = fe_mulw(aa, ML_C);                               // 39081 AA
    const fe e = fe_weak(fe_sub<2>(aa, bb));                        // E = AA - BB            mag 1
    x2 = fe_mul(caa, bb);                                           // x2 = 39081 AA BB
    z2 = fe_mul(fe_add(caa, e), e);                                 // z2 = E (39081 AA + E)  (2 x 1)
}

// The same step on the signed, register-paired limbs of gf28s.hpp: no bias and no weak reduction behind a difference,
// the sums of products added pair-wise.  The state stays in that form for all 446 steps.
struct MlStateS {
    sfp x2, z2, x3, z3;   // products: mag 1, pairable
};
GD_FN void ml_step_sel_s(MlStateS &st, const smultiplier &m1, bool sw) {
    const sfp s2 = sfe_add(st.x2, st.z2), s3 = sfe_add(st.x3, st.z3);          // mag 2, pairable
    const sfs d2 = sfe_sub(st.x2, st.z2), d3 = sfe_sub(st.x3, st.z3);          // mag 1, signed
    const sfp da = sfe_mul(s2, d3);                                             // DA (or CB, swapped)     2 x 1
    const sfp cb = sfe_mul(s3, d2);
    const sfp t1 = sfe_select(s2, s3, sw);                                      // A = sum of the pair to double
    const sfs t2 = sfe_select(d2, d3, sw);                                      // B = its difference
    st.z3 = sfe_mul(sfe_sqr<false>(sfe_sub(da, cb)), m1);                       // z3 = x1 (DA - CB)^2
    st.x3 = sfe_sqr<true>(sfe_add(da, cb));                                     // x3 = (DA + CB)^2        (sum of two products)
    const sfp aa = sfe_sqr<true>(t1);                                           // AA                      (sum of two products)
    const sfp bb = sfe_sqr<false>(t2);                                          // BB
    const sfp caa = sfe_mulw(aa, (int32_t)ML_C);                                // 39081 AA
    const sfs e = sfe_sub(aa, bb);                                              // E = AA - BB             mag 1, signed
    st.x2 = sfe_mul(caa, bb);                                                   // x2 = 39081 AA BB
    st.z2 = sfe_mul(sfe_add(caa, e), e);                                        // z2 = E (39081 AA + E)   2 x 1
}

// R = the point with u(R) = X1 / Z1, given u(R + P) = X2 / Z2 and P = b with x1 = u(P) in affine form: the y-coordinate by
// Okeya and Sakurai's formula, the result in extended coordinates (src/goldilocks.c has no counterpart: the reference
// multiplies with tables).  Exceptional cases by selects: R trivial (the identity or (0, -1)) or P trivial: the identity;
// R + P trivial: R = -P.
GD_FN pt ml_recover(const pt &b, const fe &x1, const fe &X1, const fe &Z1, const fe &X2, const fe &Z2) {
    const fe yz = fe_add(b.y, b.z);                 // Y + Z                  mag 2
    const bool q_trivial = fe_is_zero(X1) | fe_is_zero(Z1);     // '|', not '||': no branch on the scalar
    const bool qp_trivial = fe_is_zero(X2) | fe_is_zero(Z2);
    const bool base_trivial = fe_is_zero(b.x);

    // Okeya-Sakurai: with x = u(P), y = v(P) = vn / vd,
    //   39081 * Ynum = Z2 [(X1 x + Z1)(39081 (X1 + x Z1) + 156332 Z1) - 156332 Z1^2] - 39081 (X1 - x Z1)^2 X2
    //   (U : V : W) = (-8 vn X1 Z1 Z2 : 39081 Ynum vd : -8 vn Z1^2 Z2),   vn = -(Y+Z) Z,  vd = (Y-Z) X
    // and x_E = -U/V, y_E = (U + W)/(U - W).  Signs are arranged so that nothing is negated.
    const fe xz1 = fe_mul(x1, Z1);
    const fe t1 = fe_add(fe_mul(X1, x1), Z1);                                   // mag 2
    const fe t2 = fe_add(fe_mulw(fe_add(X1, xz1), ML_C), fe_mulw(Z1, ML_2AC));  // mag 2
    const fe t3 = fe_weak(fe_sub<2>(fe_mul(t1, t2), fe_mulw(fe_sqr(Z1), ML_2AC)));
    const fe t4 = fe_weak(fe_sub<2>(X1, xz1));
    const fe q4 = fe_mulw(fe_mul(fe_sqr(t4), X2), ML_C);
    const fe vneg = fe_mul(fe_weak(fe_sub<2>(q4, fe_mul(Z2, t3))),              // -39081 Ynum ...
                           fe_mul(ml_denominator(b), b.x));                      // ... times vd
    const fe kp = fe_mulw(fe_mul(fe_mul(yz, b.z), fe_mul(Z1, Z2)), 8);          // 8 (Y+Z) Z Z1 Z2
    const fe kx = fe_mul(kp, X1);
    const fe m = fe_weak(fe_sub<2>(X1, Z1));
    const fe pl = fe_add(X1, Z1);                                               // mag 2
    pt r;
    r.x = fe_mul(kx, m);
    r.y = fe_mul(vneg, pl);
    r.z = fe_mul(vneg, m);
    r.t = fe_mul(kx, pl);
    // (s+1) P trivial: s P = -P
    const pt nb = pt_negate(b);
    r.x = fe_select(r.x, nb.x, qp_trivial);
    r.y = fe_select(r.y, nb.y, qp_trivial);
    r.z = fe_select(r.z, nb.z, qp_trivial);
    r.t = fe_select(r.t, nb.t, qp_trivial);
    const bool ident = q_trivial | base_trivial;
    const pt id = pt_identity();
    r.x = fe_select(r.x, id.x, ident);
    r.y = fe_select(r.y, id.y, ident);
    r.z = fe_select(r.z, id.z, ident);
    r.t = fe_select(r.t, id.t, ident);
    return r;
}

// bits.word(k): k-th 32-bit word of the scalar, already reduced mod q (446 bits).
// b: the base point; x1 = u(P) = (Y + Z)/(Y - Z) in affine form (anything if P is the identity or (0,-1)).
template <class BITS>
GD_FN pt ml_scalarmul_u(const pt &b, const fe &x1, const BITS &bits) {
    // the ladder's state lives in the signed, register-paired form of gf28s.hpp for all 446 steps
    MlStateS st;
    st.x2 = sfe_from_fe(fe_one());
    st.z2 = sfe_from_fe(fe_zero());
    st.x3 = sfe_from_fe(x1);
    st.z3 = sfe_from_fe(fe_one());
    const smultiplier m1 = s_multiplier(st.x3);     // x1's half sums are loop-invariant
    bool swap = false;
    // one read of the scalar per 32 steps: the word's next bit is kept in the sign position
#pragma unroll 1
    for (int wi = (ML_BITS - 1) >> 5; wi >= 0; wi--) {
        const int top = wi == ((ML_BITS - 1) >> 5) ? ((ML_BITS - 1) & 31) : 31;
        uint32_t w = bits.word(wi) << (31 - top);
#pragma unroll 1
        for (int j = top; j >= 0; j--) {
            const bool k_t = (int32_t)w < 0;
            w <<= 1;
            const bool sw = swap != k_t;
            swap = k_t;
            ml_step_sel_s(st, m1, sw);
        }
    }
    // (X1 : Z1) = u(sP), (X2 : Z2) = u((s+1)P), back in the unsigned form (mag 1)
    const fe x2 = sfe_to_fe(st.x2), z2 = sfe_to_fe(st.z2), x3 = sfe_to_fe(st.x3), z3 = sfe_to_fe(st.z3);
    const fe X1 = fe_select(x2, x3, swap), Z1 = fe_select(z2, z3, swap);
    const fe X2 = fe_select(x3, x2, swap), Z2 = fe_select(z3, z2, swap);
    return ml_recover(b, x1, X1, Z1, X2, Z2);
}
// ... given di = 1/(Y - Z) (anything if Y = Z): what a kernel that shares its inversions has at hand
template <class BITS>
GD_FN pt ml_scalarmul(const pt &b, const fe &di, const BITS &bits) {
    return ml_scalarmul_u(b, fe_mul(fe_add(b.y, b.z), di), bits);
}

}  // namespace gd
