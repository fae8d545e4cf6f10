// x448.hpp -- RFC 7748 X448 on the same lane arithmetic ("next" row f3 of SURVEY.md section 8):
//   goldilocks_x448                       src/goldilocks.c:1006-1076   Montgomery ladder, 448 steps of
//                                                                      5M + 4S + 1 mulw
//   goldilocks_x448_derive_public_key     src/goldilocks.c:1115-1141   fixed-base comb + (y/x)^2
#pragma once
#include "scalarmul.hpp"
#include "gf28s.hpp"

namespace gd {

// BITS: bits.word(k) = k-th 32-bit word of the 56-byte scalar as given (clamping is applied here).
// base/out: 56-byte strings as 14 words.  Returns false iff the result is zero (reference :1065,1075).
// x448_ladder leaves the result as the fraction rx / rz; x448_finish takes 1/rz (0 for rz = 0, as
// gf_invert(0) = 0 in the reference) -- two halves so a kernel can share one inversion between the
// operations of a lane.
template <class BITS>
GD_FN void x448_ladder(fe &rx, fe &rz, const uint32_t base[14], const BITS &bits) {
    fe x1;
    (void)fe_deserialize_words(x1, base);   // the reference ignores the range check too (:1014)
    // the state stays in the signed, register-paired form of gf28s.hpp (no bias, no weak reduction behind a difference)
    sfp x2 = sfe_from_fe(fe_one()), z2 = sfe_from_fe(fe_zero()), x3 = sfe_from_fe(x1), z3 = sfe_from_fe(fe_one());
    const smultiplier m1 = s_multiplier(x3);
    bool swap = false;
#pragma unroll 1
    for (int t = 447; t >= 0; t--) {
        uint32_t bit = (bits.word(t >> 5) >> (t & 31)) & 1u;
        if (t < 2) bit = 0;          // scalar[0] &= -COFACTOR
        if (t == 447) bit = 1;       // top bit forced
        const bool k_t = bit != 0;
        const bool sw = swap != k_t;
        swap = k_t;
        // Exchanging the two pairs exchanges DA and CB: DA + CB and (DA - CB)^2 do not depend on the swap, only the
        // pair that is doubled is selected -- one sum and one difference, 32 selects per step, not 64
        // (montgomery.hpp ml_step_sel_s).
        const sfp s2 = sfe_add(x2, z2), s3 = sfe_add(x3, z3);          // mag 2, pairable
        const sfs d2 = sfe_sub(x2, z2), d3 = sfe_sub(x3, z3);          // mag 1, signed
        const sfp da = sfe_mul(s2, d3);                 // DA (or CB: the same pair)
        const sfp cb = sfe_mul(s3, d2);
        const sfp t1 = sfe_select(s2, s3, sw);          // A = x2 + z2 of the pair to double
        const sfs t2 = sfe_select(d2, d3, sw);          // B = x2 - z2
        z3 = sfe_mul(sfe_sqr<false>(sfe_sub(da, cb)), m1);   // z3 = x1 (DA-CB)^2
        x3 = sfe_sqr<true>(sfe_add(da, cb));            // x3 = (DA+CB)^2
        const sfp aa = sfe_sqr<true>(t1);               // AA
        const sfp bb = sfe_sqr<false>(t2);              // BB
        x2 = sfe_mul(aa, bb);
        const sfs e = sfe_sub(aa, bb);                  // E = AA - BB            mag 1, signed
        const sfp f = sfe_add(sfe_mulw(e, 39081), aa);  // AA + a24 E             mag 2
        z2 = sfe_mul(f, e);
    }
    const fe x2u = sfe_to_fe(x2), z2u = sfe_to_fe(z2), x3u = sfe_to_fe(x3), z3u = sfe_to_fe(z3);
    rx = fe_select(x2u, x3u, swap);
    rz = fe_select(z2u, z3u, swap);
}
GD_FN bool x448_finish(uint32_t out[14], const fe &rx, const fe &rzi) {
    fe r = fe_mul(rx, rzi);
    fe_serialize_words(out, r);
    return !fe_is_zero(r);
}
template <class BITS>
GD_FN bool x448_core(uint32_t out[14], const uint32_t base[14], const BITS &bits) {
    fe rx, rz;
    x448_ladder(rx, rz, base, bits);
    return x448_finish(out, rx, fe_invert(rz));
}

// (y/x)^2 of a twisted-Edwards point, serialized (src/goldilocks.c:1102-1113); xi = 1/x
GD_FN void x448_public_finish(uint32_t out[14], const fe &y, const fe &xi) {
    fe r = fe_mul(xi, y);
    fe_serialize_words(out, fe_sqr(r));
}
GD_FN void pt_encode_x448_words(uint32_t out[14], const pt &p) { x448_public_finish(out, p.y, fe_invert(p.x)); }

// the scalar x448_derive_public_key multiplies the base point by (src/goldilocks.c:1119-1136)
GD_FN sc x448_public_scalar(const uint32_t scalar_words[14]) {
    sc s;
#pragma unroll
    for (int i = 0; i < 14; i++) s.w[i] = scalar_words[i];
    s.w[0] &= ~3u;
    s.w[13] = (s.w[13] & 0x7fffffffu) | 0x80000000u;
    return sc_halve(sc_reduce(s));           // decode_long of exactly 56 bytes = reduce; ratio 2 = one halving
}

}  // namespace gd
