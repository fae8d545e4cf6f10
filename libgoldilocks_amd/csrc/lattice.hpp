// lattice.hpp -- half-size scalars for signature verification (per lane, integer arithmetic only).
//
// Verification checks  S*B - h*A == R  with a 446-bit h: one variable-base ladder of 446 doublings.
// Following Antipa, Brown, Gallant, Lambert, Struik and Vanstone ("Accelerated verification of ECDSA
// signatures", SAC 2005; for EdDSA: Pornin, "Optimized lattice basis reduction in dimension 2, and fast
// Schnorr and EdDSA signature verification", 2020) the equation is multiplied by a small nonzero tau:
//       (tau*S)*B - rho*A - tau*R == 0,        rho == tau*h  (mod q),   0 <= rho < 2^223, 0 < |tau| < 2^223,
// so that the two variable points A and R share ONE ladder of 225 doublings (45 five-bit windows; the
// base-point term costs no doublings at all here).
//
// Why the accept set is the reference's, bit for bit: every point the verification handles lies in the
// subgroup of PRIME order q of the internal twisted curve.  The decoding applies the 4-isogeny whose
// kernel is the whole rational torsion of Ed448 (src/goldilocks.c:949-1004 -- "decode_like_eddsa_and_mul_
// by_ratio", ratio 4, which src/eddsa.c:283-300 compensates by multiplying S by 4), so an encoding with a
// torsion component decodes to the same internal point as the one without (fixture F7: the reference
// accepts those signatures) and the image, of order 4q/4, is the subgroup of order q; the base point
// generates it.  There point_eq's "equal up to 2-torsion" (src/goldilocks.c:644-653) is plain equality,
// D = S*B - h*A - R is the identity iff tau*D is for any tau prime to q, and scalars act modulo q.
// (A CPU test checks q*P = 0 for every decodable point of fixtures F7 and F3, torsion-shifted and
// small-order encodings included.)
//
// The pair comes from the Euclidean remainder sequence of (q, h) with its cofactors, stopped at the first
// remainder below 2^223: r_i == t_i*h (mod q) and |t_i| <= q / r_(i-1) < 2^223.
#pragma once
#include "sc14.hpp"

namespace gd {

struct wide15 {   // unsigned, 480 bits
    uint32_t w[15];
};
struct int8w {    // signed two's complement, 256 bits
    uint32_t w[8];
};

constexpr int LATTICE_WINDOWS = 45;   // 225 bits hold both halves of a pair

GD_FN int bitlen15(const wide15 &a) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 15; i++)
        if (a.w[i]) n = 32 * i + (32 - __builtin_clz(a.w[i]));
    return n;
}
GD_FN bool at_least_2_223(const wide15 &a) {   // a >= 2^223
    uint32_t acc = a.w[6] >> 31;
#pragma unroll
    for (int i = 7; i < 15; i++) acc |= a.w[i];
    return acc != 0;
}
GD_FN void shl1(wide15 &a) {
#pragma unroll
    for (int i = 14; i > 0; i--) a.w[i] = a.w[i] << 1 | a.w[i - 1] >> 31;
    a.w[0] <<= 1;
}
GD_FN void shr1(wide15 &a) {
#pragma unroll
    for (int i = 0; i < 14; i++) a.w[i] = a.w[i] >> 1 | a.w[i + 1] << 31;
    a.w[14] >>= 1;
}
GD_FN void shl1(int8w &a) {
#pragma unroll
    for (int i = 7; i > 0; i--) a.w[i] = a.w[i] << 1 | a.w[i - 1] >> 31;
    a.w[0] <<= 1;
}
GD_FN void sar1(int8w &a) {   // arithmetic
#pragma unroll
    for (int i = 0; i < 7; i++) a.w[i] = a.w[i] >> 1 | a.w[i + 1] << 31;
    a.w[7] = (uint32_t)((int32_t)a.w[7] >> 1);
}
// if a >= b: a -= b, returns true
GD_FN bool sub_if_ge(wide15 &a, const wide15 &b) {
    wide15 d;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        c += (int64_t)a.w[i] - (int64_t)b.w[i];
        d.w[i] = (uint32_t)c;
        c >>= 32;
    }
    const bool ge = c == 0;
#pragma unroll
    for (int i = 0; i < 15; i++) a.w[i] = ge ? d.w[i] : a.w[i];
    return ge;
}
GD_FN void sub_if(int8w &a, const int8w &b, bool doit) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.w[i] - (int64_t)(doit ? b.w[i] : 0u);
        a.w[i] = (uint32_t)c;
        c >>= 32;
    }
}

// One exact step of the remainder sequence at full precision: (r0, t0) -= floor(r0 / r1) * (r1, t1) by
// shift-and-subtract, then the pairs change places.  Needs r0 >= r1 > 0.
GD_FN void euclid_step_exact(wide15 &r0, int8w &t0, wide15 &r1, int8w &t1) {
    const int d = bitlen15(r0) - bitlen15(r1);
    wide15 rs = r1;
    int8w ts = t1;
    for (int k = 0; k < d; k++) {
        shl1(rs);
        shl1(ts);
    }
    for (int k = d; k >= 0; k--) {
        const bool ge = sub_if_ge(r0, rs);
        sub_if(t0, ts, ge);
        shr1(rs);
        sar1(ts);
    }
    const wide15 rt = r0;
    r0 = r1;
    r1 = rt;
    const int8w tt = t0;
    t0 = t1;
    t1 = tt;
}

// The reciprocal behind the single-precision quotient estimate.  Its accuracy affects SPEED ONLY, never the
// result: the estimate q is corrected by one either way from its remainder, and a step is taken only if the
// remainders of BOTH bracketing quotients are in range (rem, rem2 below) -- otherwise the inner loop stops and
// the step is made exactly at full precision.  The device uses v_rcp_f64 (relative error about 2^-23 on this
// part); the host checker build can perturb its exact 1/v by a relative gd_rcp_perturb() to run the same paths.
#if !defined(__HIPCC__)
inline double &gd_rcp_perturb() {
    static double rel = 0.0;
    return rel;
}
#endif
GD_FN double fast_rcp(double v) {
#if defined(__HIPCC__)
    return __builtin_amdgcn_rcp(v);
#else
    return (1.0 / v) * (1.0 + gd_rcp_perturb());
#endif
}

// bits [sh, sh + 63) of a (sh differs from lane to lane: selects, not indexing -- the words stay in registers)
GD_FN uint64_t window63(const wide15 &a, int sh) {
    const int ws = sh >> 5, bs = sh & 31;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        w0 = i == ws ? a.w[i] : w0;
        w1 = i == ws + 1 ? a.w[i] : w1;
        w2 = i == ws + 2 ? a.w[i] : w2;
    }
    const uint64_t lo = (uint64_t)w1 << 32 | w0;
    const uint64_t v = bs ? lo >> bs | (uint64_t)w2 << (64 - bs) : lo;
    return v & 0x7fffffffffffffffull;
}

// x <- +-(a*x - b*y) over N words, two's complement modulo 2^(32 N); a, b < 2^30.
// The caller knows the sign of the exact value (`negate`: the value a*x - b*y is <= 0 and its negation is wanted).
template <int N>
GD_FN void lincomb_words(uint32_t (&out)[N], const uint32_t (&x)[N], uint32_t a, const uint32_t (&y)[N], uint32_t b,
                         bool negate) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (int64_t)((uint64_t)a * x[i]) - (int64_t)((uint64_t)b * y[i]);   // |.| < 2^62 + carry
        out[i] = (uint32_t)c;
        c >>= 32;
    }
    uint64_t n = 1;
#pragma unroll
    for (int i = 0; i < N; i++) {
        n += (uint32_t)~out[i];
        out[i] = negate ? (uint32_t)n : out[i];
        n >>= 32;
    }
}

// The short pair for challenge h (< q):  0 <= rho < 2^223, 0 < |tau| < 2^223, rho == tau * h (mod q).
//
// Lehmer's method (Knuth, TAOCP vol. 2, 4.5.2, Algorithm L): the quotients of the remainder sequence are
// found on the leading 63 bits of (r0, r1) -- a quotient is taken only while the two bracketing single-
// precision quotients agree, which makes it the exact one -- and about 30 bits' worth of steps are then
// applied to the long numbers as one 2x2 matrix with entries below 2^30.  The single-precision run also
// stops short of the 2^223 line (the threshold T leaves room for the truncation error), so the crossing
// itself is always an exact full-precision step and the stopping rule is the one stated above.
// The host checker compares the pair with the remainder sequence computed in Python integers.
GD_FN void half_size_pair(wide15 &rho, int8w &tau, const sc &h) {
    wide15 r0, r1;
    int8w t0, t1;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        r0.w[i] = i < 14 ? SC_Q[i] : 0u;
        r1.w[i] = i < 14 ? h.w[i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t0.w[i] = 0;
        t1.w[i] = i == 0 ? 1u : 0u;
    }
    while (at_least_2_223(r1)) {
        const int sh = bitlen15(r0) - 63;                                // >= 161
        uint64_t x = window63(r0, sh), y = window63(r1, sh);
        const uint64_t T = (1ull << 31) + (sh <= 223 ? 1ull << (223 - sh) : 0ull);
        int32_t A = 1, B = 0, C = 0, D = 1;                              // |.| < 2^30
        for (int it = 0; it < 48; it++) {
            // one candidate step, computed unconditionally; `go` collects what the exactness argument needs
            const uint64_t yc = y + (uint64_t)(int64_t)C, yd = y + (uint64_t)(int64_t)D;
            const uint64_t xa = x + (uint64_t)(int64_t)A, xb = x + (uint64_t)(int64_t)B;
            bool go = yc != 0 && yd != 0;
            // q = floor(xa / yc) when below 2^20: a double-precision estimate, then made exact by its remainder
            const double qd = (double)xa * fast_rcp((double)yc);
            go = go && qd < 1048576.0;                                     // false for NaN / infinity too
            uint32_t q = go ? (uint32_t)qd : 0u;
            int64_t rem = (int64_t)(xa - (uint64_t)q * yc);
            const bool under = rem < 0;
            q -= under ? 1u : 0u;
            rem += under ? (int64_t)yc : 0;
            const bool over = rem >= (int64_t)yc;
            q += over ? 1u : 0u;
            rem -= over ? (int64_t)yc : 0;
            go = go && rem >= 0 && rem < (int64_t)yc;                      // always: the estimate is within one
            const int64_t rem2 = (int64_t)(xb - (uint64_t)q * yd);        // q == floor(xb / yd) ?   (no overflow: q*yd < 2^64)
            go = go && rem2 >= 0 && rem2 < (int64_t)yd;
            const uint64_t ny = x - (uint64_t)q * y;
            go = go && ny >= T;
            const int64_t nC = (int64_t)A - (int64_t)q * C, nD = (int64_t)B - (int64_t)q * D;
            const int64_t lim = 1ll << 30;
            go = go && nC < lim && nC > -lim && nD < lim && nD > -lim;
            if (!go) break;
            A = C;
            C = (int32_t)nC;
            B = D;
            D = (int32_t)nD;
            x = y;
            y = ny;
        }
        if (B == 0) {
            euclid_step_exact(r0, t0, r1, t1);
        } else {
            // an even number of steps: A > 0 >= B, C < 0 < D; an odd number: A <= 0 < B, C > 0 >= D
            const bool odd = A <= 0;
            const uint32_t a = (uint32_t)(A < 0 ? -A : A), b = (uint32_t)(B < 0 ? -B : B);
            const uint32_t c = (uint32_t)(C < 0 ? -C : C), d = (uint32_t)(D < 0 ? -D : D);
            wide15 n0, n1;
            int8w u0, u1;
            lincomb_words<15>(n0.w, r0.w, a, r1.w, b, odd);              // r0' = A r0 + B r1
            lincomb_words<15>(n1.w, r1.w, d, r0.w, c, odd);              // r1' = C r0 + D r1
            lincomb_words<8>(u0.w, t0.w, a, t1.w, b, odd);
            lincomb_words<8>(u1.w, t1.w, d, t0.w, c, odd);
            r0 = n0;
            r1 = n1;
            t0 = u0;
            t1 = u1;
        }
    }
    rho = r1;
    tau = t1;
}

GD_FN bool is_negative(const int8w &a) { return (int32_t)a.w[7] < 0; }
GD_FN sc magnitude_as_scalar(const int8w &a) {   // |a| < 2^223, as a scalar
    const bool neg = is_negative(a);
    sc s = sc_zero();
    uint64_t c = neg ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += neg ? (uint32_t)~a.w[i] : a.w[i];
        s.w[i] = (uint32_t)c;
        c >>= 32;
    }
    return s;
}

// Signed 5-bit windows of an ODD positive integer s < 2^(5 nw): the words of s' = (s + 2^(5 nw) - 1) / 2, whose
// window digits w_i give s = sum (2 w_i - 31) 32^i (src/goldilocks.c:420-438 without the reduction mod q,
// which would bring all 446 bits back).  Everything but the top bit: s >> 1; the caller sets bit 5 nw - 1.
GD_FN void recode_odd_base(uint32_t out[15], const wide15 &s) {
#pragma unroll
    for (int i = 0; i < 14; i++) out[i] = s.w[i] >> 1 | s.w[i + 1] << 31;
    out[14] = s.w[14] >> 1;
}

}  // namespace gd
