// point.hpp -- extended twisted Edwards arithmetic on E: -x^2 + y^2 = 1 + d' x^2 y^2,
// d' = -39082 (the curve the reference computes on, src/goldilocks.c:32-47), one point
// per lane, coordinates as gd::fe (16 x 28-bit limbs in VGPRs).
//
// Formulas restate the reference's (SURVEY.md section 9 gives them as math):
//   doubling          src/goldilocks.c:232-254   4S + 3M (+1M for T)
//   mixed addition    src/goldilocks.c:314-380   6M (+1M for Z*z, +1M for T)
//   full addition     src/goldilocks.c:205-230
//   niels conversions src/goldilocks.c:271-312
// but with their own lazy-reduction schedule for 28-bit limbs (see the magnitude
// contract in gf28.hpp): every product below has mag(a)*mag(b) <= 6 (limit 6.7).
//
// Sign convention: our (p)niels keep cn = -c_ref = +2*39082*T (so no negation is
// needed when a table entry is built); entries imported from reference-format
// tables are negated once when they are staged.
#pragma once
#include "gf28.hpp"

namespace gd {

struct pt {
    fe x, y, z, t;
};
struct pniels {  // projective niels: a = Y-X, b = Y+X, cn = 2*39082*T, z = 2Z
    fe a, b, cn, z;
};
struct niels {  // affine niels (Z = 1)
    fe a, b, cn;
};

constexpr uint32_t TWO_EFF_D = 2 * 39082;  // 2 * (-d')
constexpr uint32_t FOUR_EFF_D = 4 * 39082; // 156328 = -4 d'
constexpr uint32_t NEG_EDWARDS_D = 39081;  // -d  (also 1 - (-d') .. "-1-TWISTED_D")

GD_FN pt pt_identity() {
    pt p;
    p.x = fe_zero();
    p.y = fe_one();
    p.z = fe_one();
    p.t = fe_zero();
    return p;
}

// P <- 2P.  Inputs x,y,z mag 1 (+eps); outputs mag 1.  T is produced only when
// need_t (the reference's !before_double); need_t must be wave-uniform.
GD_FN void pt_double(pt &p, bool need_t) {
    fe c = fe_sqr(p.x);
    fe a = fe_sqr(p.y);
    fe d = fe_add(c, a);                         // X^2 + Y^2                     mag 2
    fe s = fe_add(p.x, p.y);                     //                               mag 2
    fe b = fe_sub<3>(fe_sqr(s), d);              // (X+Y)^2 - X^2 - Y^2           mag 4
    fe tt = fe_sub<2>(a, c);                     // Y^2 - X^2                     mag 3
    fe zz = fe_sqr(p.z);
    fe e = fe_weak(fe_sub<4>(fe_add(zz, zz), tt));   // 2Z^2 - (Y^2 - X^2)        mag 1
    // one weak reduction covers the three products every doubling needs (1x4, 1x3, 2x3 <= 6.7)
    p.x = fe_mul(e, b);
    p.z = fe_mul(e, tt);
    p.y = fe_mul(d, tt);
    if (need_t) p.t = fe_mul(d, fe_weak(b));     // 2 x 4 would not fit: reduce b for this one
}

// Core of the mixed additions.  zz = Z (niels) or Z*z (pniels); ea/eb/cn are the
// entry's fields; neg subtracts the entry instead of adding it.
// need_t must be wave-uniform (it is a branch, not a select).
GD_FN void pt_add_core(pt &p, const fe &zz, const fe &ea_, const fe &eb_, const fe &cn, bool neg,
                       bool need_t) {
    fe u = fe_sub<2>(p.y, p.x);                  // mag 3
    fe v = fe_add(p.x, p.y);                     // mag 2
    fe ea = fe_select(ea_, eb_, neg);
    fe eb = fe_select(eb_, ea_, neg);
    fe A = fe_mul(u, ea);
    fe B = fe_mul(v, eb);
    fe Cn = fe_mul(cn, p.t);                     // = -C of the reference
    fe E = fe_weak(fe_sub<2>(B, A));             // mag 1
    fe H = fe_add(A, B);                         // mag 2
    fe zm = fe_sub<2>(zz, Cn);                   // Z - Cn = Z + C   (G when adding)  mag 3: times E (1), H or zp (2)
    fe zp = fe_add(zz, Cn);                      // Z + Cn = Z - C   (F when adding)  mag 2
    fe F = fe_select(zp, zm, neg);
    fe G = fe_select(zm, zp, neg);
    p.x = fe_mul(F, E);
    p.y = fe_mul(G, H);
    p.z = fe_mul(F, G);
    if (need_t) p.t = fe_mul(E, H);
}

GD_FN void pt_add_pniels(pt &p, const pniels &e, bool neg, bool need_t) {
    fe zz = fe_mul(e.z, p.z);                    // e.z mag 2
    pt_add_core(p, zz, e.a, e.b, e.cn, neg, need_t);
}
GD_FN void pt_add_niels(pt &p, const niels &e, bool neg, bool need_t) {
    pt_add_core(p, p.z, e.a, e.b, e.cn, neg, need_t);
}

// point -> projective niels (src/goldilocks.c:280-288).  Input mag 1; a,b,cn mag 1, z mag 2.
GD_FN pniels pt_to_pniels(const pt &p) {
    pniels e;
    e.a = fe_weak(fe_sub<2>(p.y, p.x));
    e.b = fe_weak(fe_add(p.x, p.y));
    e.cn = fe_mulw(p.t, TWO_EFF_D);
    e.z = fe_add(p.z, p.z);
    return e;
}

// (+-) projective niels -> point (src/goldilocks.c:290-301)
GD_FN pt pniels_to_pt(const pniels &e, bool neg) {
    pt p;
    fe eu = fe_add(e.a, e.b);                                         // 2Y   mag 2
    fe ym = fe_weak(fe_select(fe_sub<2>(e.b, e.a), fe_sub<2>(e.a, e.b), neg));  // +-2X mag 1
    p.t = fe_mul(eu, ym);
    p.x = fe_mul(e.z, ym);
    p.y = fe_mul(e.z, eu);
    p.z = fe_sqr(e.z);
    return p;
}
// (+-) affine niels -> point (src/goldilocks.c:303-312)
GD_FN pt niels_to_pt(const niels &e, bool neg) {
    pt p;
    p.y = fe_weak(fe_add(e.a, e.b));
    p.x = fe_weak(fe_select(fe_sub<2>(e.b, e.a), fe_sub<2>(e.a, e.b), neg));
    p.t = fe_mul(p.y, p.x);
    p.z = fe_one();
    return p;
}

// Full addition / subtraction of extended points (src/goldilocks.c:178-230).  Inputs mag 1.
GD_FN pt pt_add(const pt &q, const pt &r, bool subtract) {
    pniels e = pt_to_pniels(r);
    pt p = q;
    pt_add_pniels(p, e, subtract, true);
    return p;
}

GD_FN pt pt_negate(const pt &q) {
    pt p;
    p.x = fe_weak(fe_neg(q.x));
    p.y = q.y;
    p.z = q.z;
    p.t = fe_weak(fe_neg(q.t));
    return p;
}

// Equality in the quotient by 2-torsion: X1*Y2 == Y1*X2 (src/goldilocks.c:644-653).
GD_FN bool pt_eq(const pt &p, const pt &q) { return fe_eq(fe_mul(p.y, q.x), fe_mul(q.y, p.x)); }

// Curve + extended-coordinate consistency check (src/goldilocks.c:655-673).
GD_FN bool pt_valid(const pt &p) {
    bool ok = fe_eq(fe_mul(p.x, p.y), fe_mul(p.z, p.t));
    fe lhs = fe_weak(fe_sub<2>(fe_sqr(p.y), fe_sqr(p.x)));          // y^2 - x^2
    fe dt2 = fe_mulw(fe_sqr(p.t), 39082);                            // -d' t^2
    fe rhs = fe_weak(fe_sub<2>(fe_sqr(p.z), dt2));                   // z^2 + d' t^2
    ok = ok && fe_eq(lhs, rhs);
    ok = ok && !fe_is_zero(p.z);
    return ok;
}

// 1/sqrt(39082/39081 - 1) (src/goldilocks.c:41-43 GOLDILOCKS_448_FACTOR), 28-bit limbs.
GD_CONST uint32_t FACTOR28[16] = {0x5572736u, 0x42ef0f4u, 0x0ce5296u, 0x7bf6aa2u, 0xed26033u, 0xf4fd6edu,
                                  0xa839a66u, 0x968c14bu, 0x4a2d780u, 0xb8d54b6u, 0x1a7b8a5u, 0x6aa0a1fu,
                                  0xd722fa2u, 0x683bf68u, 0xbeb24f7u, 0x22d962fu};
GD_FN fe fe_factor() {
    fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.v[i] = FACTOR28[i];
    return r;
}

// Decaf encoding: canonical 56-byte string as 14 words (src/goldilocks.c:98-140, toggles 0).
GD_FN void pt_encode_words(uint32_t out[14], const pt &p) {
    fe num = fe_mul(fe_add(p.x, p.t), fe_weak(fe_sub<2>(p.x, p.t)));   // (X+T)(X-T)
    fe x2 = fe_sqr(p.x);
    fe t2 = fe_mulw(fe_mul(x2, num), NEG_EDWARDS_D);                   // 39081 X^2 num
    bool ok;
    fe r = fe_isr(t2, &ok);
    fe ratio = fe_mul(r, num);
    bool negx = fe_lobit(fe_mul(ratio, fe_factor()));
    ratio = fe_weak(fe_cond_neg(ratio, negx));
    fe t3 = fe_weak(fe_sub<2>(fe_mul(ratio, p.z), p.t));
    fe t4 = fe_mulw(fe_mul(t3, p.x), NEG_EDWARDS_D);
    fe s = fe_mul(t4, r);
    s = fe_cond_neg(s, fe_lobit(s));
    fe_serialize_words(out, s);
}

// Decaf decoding (src/goldilocks.c:142-176).  Returns success; p is always a valid
// point representation arithmetic-wise, meaningful only on success.
GD_FN bool pt_decode_words(pt &p, const uint32_t in[14], bool allow_identity) {
    fe s;
    bool ok = fe_deserialize_words(s, in);
    ok = ok && (allow_identity || !fe_is_zero(s));
    ok = ok && !fe_lobit(s);
    fe s2 = fe_sqr(s);
    fe den = fe_weak(fe_sub<2>(fe_one(), s2));          // 1 - s^2
    fe ynum = fe_add(fe_one(), s2);                     // 1 + s^2   mag 2
    fe den2 = fe_sqr(den);
    fe num = fe_weak(fe_add(den2, fe_mulw(s2, FOUR_EFF_D)));  // den^2 - 4 d' s^2
    bool sq;
    fe isr = fe_isr(fe_mul(num, den2), &sq);
    ok = ok && sq;
    fe tmp = fe_mul(isr, den);
    p.y = fe_mul(tmp, ynum);
    fe w = fe_mul(tmp, s);
    w = fe_add(w, w);                                   // 2 s isr den   mag 2
    p.x = fe_mul(fe_mul(w, isr), num);
    p.x = fe_weak(fe_cond_neg(p.x, fe_lobit(fe_mul(w, fe_factor()))));
    p.z = fe_one();
    p.t = fe_mul(p.x, p.y);
    return ok;
}

// -1/156324 mod p = -1/(4 d' + 4) ... the constant of u(P) below
GD_CONST uint32_t NEG_INV_156324[16] = {0x157aa51u, 0xf0cf349u, 0x1278ef6u, 0xd9a3e9cu, 0x0594bdfu, 0xdfb66ccu,
                                        0x618ccd8u, 0x2fb57b4u, 0xfd93896u, 0x56bc321u, 0xe7aa36fu, 0x174cd4au,
                                        0x1cf5b23u, 0x39836c3u, 0x8e03c94u, 0xca1d2d0u};
// Decaf decoding that also hands out u(P) = (y + 1)/(y - 1), the affine Montgomery coordinate the table-free
// ladder starts from (montgomery.hpp), WITHOUT a second exponentiation.  With m = num den^2 the decoder's root
// is isr(m); here j = isr(m s^4) = isr(m)/s^2 exactly (the exponentiation is multiplicative and s^(p-3) = s^-2),
// so isr(m) = j s^2 -- the reference's value, bit for bit -- and 1/s^2 = j isr(m) m comes with it.  Then
//     y = ynum/sqrt(num), sqrt(num) = num tmp:   u = (ynum + num tmp)^2 / (ynum^2 - num),   ynum^2 - num = -156324 s^2.
// s = 0 (the identity, if allowed): u = 0, and the ladder's own test on X = 0 takes over.
GD_FN bool pt_decode_words_u(pt &p, fe &u, const uint32_t in[14], bool allow_identity) {
    fe s;
    bool ok = fe_deserialize_words(s, in);
    ok = ok && (allow_identity || !fe_is_zero(s));
    ok = ok && !fe_lobit(s);
    fe s2 = fe_sqr(s);
    fe den = fe_weak(fe_sub<2>(fe_one(), s2));          // 1 - s^2
    fe ynum = fe_add(fe_one(), s2);                     // 1 + s^2   mag 2
    fe den2 = fe_sqr(den);
    fe num = fe_weak(fe_add(den2, fe_mulw(s2, FOUR_EFF_D)));  // den^2 - 4 d' s^2
    const fe m = fe_mul(num, den2);
    bool sq;
    const fe j = fe_isr(fe_mul(m, fe_sqr(s2)), &sq);
    sq = sq || fe_is_zero(s);                           // s = 0: m = 1 is a square, m s^4 = 0 says nothing
    const fe isr = fe_select(fe_mul(j, s2), fe_one(), fe_is_zero(s));   // isr(1) = 1
    ok = ok && sq;
    const fe inv_s2 = fe_mul(fe_mul(j, isr), m);        // 1/s^2  (0 for s = 0)
    fe tmp = fe_mul(isr, den);
    p.y = fe_mul(tmp, ynum);
    fe w = fe_mul(tmp, s);
    w = fe_add(w, w);                                   // 2 s isr den   mag 2
    p.x = fe_mul(fe_mul(w, isr), num);
    p.x = fe_weak(fe_cond_neg(p.x, fe_lobit(fe_mul(w, fe_factor()))));
    p.z = fe_one();
    p.t = fe_mul(p.x, p.y);
    const fe top = fe_sqr(fe_weak(fe_add(ynum, fe_mul(num, tmp))));    // (ynum + sqrt(num))^2
    fe k;
#pragma unroll
    for (int i = 0; i < 16; i++) k.v[i] = NEG_INV_156324[i];
    u = fe_mul(fe_mul(top, inv_s2), k);
    return ok;
}

// RFC 8032 decoding followed by the 4-isogeny onto the twisted curve
// (src/goldilocks.c:949-1004).  in: 57 bytes as 15 words (top 3 bytes of word 14 unused).
GD_FN bool pt_decode_eddsa_words(pt &p, const uint32_t in[15]) {
    uint32_t last = in[14] & 0xff;
    bool low = (last & 0x80) != 0;
    bool ok = (last & 0x7f) == 0;
    fe y;
    ok = fe_deserialize_words(y, in) && ok;
    fe y2 = fe_sqr(y);
    fe num = fe_weak(fe_sub<2>(fe_one(), y2));                           // 1 - y^2
    fe den = fe_weak(fe_add(fe_one(), fe_mulw(y2, NEG_EDWARDS_D)));      // 1 - d y^2, d = -39081
    bool sq;
    fe isr = fe_isr(fe_mul(num, den), &sq);
    ok = ok && sq;
    fe x = fe_mul(isr, num);
    x = fe_weak(fe_cond_neg(x, fe_lobit(x) != low));
    // isogeny: like doubling with Z = 1 but E = 2 - D (not 2 - T')
    fe c = fe_sqr(x);
    fe a = y2;
    fe d = fe_add(c, a);                                                  // mag 2
    fe b = fe_weak(fe_sub<4>(fe_sqr(fe_add(x, y)), d));
    fe tt = fe_weak(fe_sub<2>(a, c));
    fe e = fe_weak(fe_sub<4>(fe_small(2), d));                            // 2 - D
    p.x = fe_mul(e, b);
    p.z = fe_mul(tt, e);
    p.y = fe_mul(d, tt);
    p.t = fe_mul(d, b);
    return ok;
}

// Elligator 2 hash-to-curve, one 56-byte string -> point ("next" row f4; src/elligator.c:32-83).
GD_FN pt pt_from_hash_words(const uint32_t in[14]) {
    fe r0;
    (void)fe_deserialize_words(r0, in);                  // any 448-bit string (reduced mod p below)
    r0 = fe_strong(r0);
    fe r = fe_weak(fe_neg(fe_sqr(r0)));                  // r = -r0^2 (qnr = -1)
    fe rm1 = fe_weak(fe_sub<2>(r, fe_one()));            // r - 1
    fe drd = fe_weak(fe_neg(fe_mulw(rm1, NEG_EDWARDS_D)));   // d r - d, d = -39081
    fe a = fe_add(drd, fe_one());                        // mag 2
    fe b = fe_weak(fe_sub<2>(drd, r));
    fe D = fe_mul(a, b);                                 // (dr - d + 1)(dr - d - r)
    fe N = fe_mulw(fe_add(r, fe_one()), 78163);          // (r + 1)(1 - 2d)
    bool square;
    fe isr = fe_isr(fe_mul(D, N), &square);
    fe e = fe_mul(isr, fe_select(r0, fe_one(), square));
    fe s = fe_mul(N, e);
    s = fe_weak(fe_cond_neg(s, fe_lobit(s) != !square)); // negate iff lobit(s) ^ ~square
    fe c = fe_mulw(e, 78163);
    fe t = fe_mul(fe_mul(fe_sqr(c), rm1), N);
    t = fe_weak(fe_cond_neg(t, square));
    t = fe_weak(fe_sub<2>(t, fe_one()));
    fe s2 = fe_sqr(s);
    fe two_s = fe_add(s, s);                             // mag 2
    fe ep = fe_add(s2, fe_one());                        // 1 + s^2, mag 2
    fe em = fe_weak(fe_sub<2>(fe_one(), s2));            // 1 - s^2
    pt p;
    p.t = fe_mul(two_s, ep);
    p.x = fe_mul(two_s, t);
    p.y = fe_mul(ep, em);
    p.z = fe_mul(em, t);
    return p;
}

// Dual isogeny back to Ed448 and RFC 8032 encoding (src/goldilocks.c:905-946), in two halves so that
// a kernel can put one shared inversion between them (Montgomery's trick along a lane's operations).
// (xn : yn : zn) = 4 * P in projective Ed448 coordinates.
GD_FN void pt_eddsa_isogeny(fe &xn, fe &yn, fe &zn, const pt &p) {
    fe x = fe_sqr(p.x);
    fe t = fe_sqr(p.y);
    fe u = fe_add(x, t);                                     // mag 2
    fe y = fe_weak(fe_sub<4>(fe_sqr(fe_add(p.x, p.y)), u));  // 2XY
    fe z = fe_weak(fe_sub<2>(t, x));                         // Y^2 - X^2
    fe zz = fe_sqr(p.z);
    fe tt = fe_weak(fe_sub<2>(fe_add(zz, zz), z));           // 2Z^2 - (Y^2 - X^2)
    xn = fe_mul(tt, y);
    yn = fe_mul(u, z);
    zn = fe_mul(u, tt);
}
// out: 57 bytes as 15 words (upper 3 bytes of word 14 are zero); zi = 1/zn
GD_FN void eddsa_finish_words(uint32_t out[15], const fe &xn, const fe &yn, const fe &zi) {
    fe xa = fe_mul(xn, zi);
    fe ya = fe_mul(yn, zi);
    fe_serialize_words(out, ya);
    out[14] = fe_lobit(xa) ? 0x80u : 0u;
}
GD_FN void pt_encode_eddsa_words(uint32_t out[15], const pt &p) {
    fe xn, yn, zn;
    pt_eddsa_isogeny(xn, yn, zn, p);
    eddsa_finish_words(out, xn, yn, fe_invert(zn));
}

}  // namespace gd
