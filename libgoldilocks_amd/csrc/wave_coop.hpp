// wave_coop.hpp -- ONE Ed448 operation per WAVEFRONT: the small-batch / single-call path.
//
// The lane-per-operation kernels need 2^16 and more operations to fill the chip, and one lane's
// ladder takes 2.1 ms however few operations there are (445 doublings x 7 dependent field
// multiplications of 274 instructions each).  Here the 64 lanes of a wave share one operation:
//
//   * a field element is spread over the 16 lanes of a ROW, lane i holding its 28-bit limb i in ONE
//     VGPR; the four rows of the wave hold four DIFFERENT field elements, so a register is a 4-vector
//     of field elements and every instruction works on four of them at once (the arrangement of the
//     AVX2 curve implementations, with a row where they have a 64-bit SIMD lane);
//   * a multiplication gives lane l column l of the product: 8 steps of 2 MACs, the multiplier's
//     limbs broadcast inside the row (ds_bpermute_b32), the multiplicand rotated inside the row
//     (DPP row_ror) -- the rotation carries the phi^2 = phi + 1 fold of p = 2^448 - 2^224 - 1 by
//     rotating pre-added halves (b0 | b1, b0+b1 | b1, b1 | b0+b1, b0+2*b1 | b0+b1) chosen per consumer
//     half with the DPP bank mask; 16 + 48 + 18 (carries) instructions instead of 274, and four
//     independent products per instruction stream;
//   * a point (X, Y, Z, T) is ONE register (rows 0..3); a doubling is two vector multiplications
//     (X^2, Y^2, Z^2, (X+Y)^2), then (E*B, D*T', E*T', D*B)), a mixed addition two more
//     ((Y-X)*a, (Y+X)*b, T*cn, Z*z), then (F*E, G*H, F*G, E*H)); rows talk through ds_bpermute_b32;
//   * the 16-entry window table lives in LDS (4 KiB per operation) and every lookup reads all 16
//     entries (index-independent: 16 ds_read_b32 + 16 v_cndmask per lane), whatever the table mode.
//
// Formulas and window recoding restate the same reference lines as point.hpp / scalarmul.hpp
// (src/goldilocks.c:232-254, :314-380, :405-465); the magnitude contract is gf28.hpp's (the column
// sums are the same 38 products at most).  Device-only: no host build of this header exists.
#pragma once
#include <hip/hip_runtime.h>

#include "abi.hpp"
#include "eddsa.hpp"
#include "scalarmul.hpp"

namespace gd {
namespace wc {

using wfe = uint32_t;   // limb (lane & 15) of the field element held by row (lane >> 4) & 3

struct Lane {           // per-lane constants, computed once per kernel
    uint32_t i;         // limb index 0..15
    uint32_t row;       // 0..3
    int i4;             // 4 * i: bpermute byte address of limb i in row 0
    int row4;           // 4 * (first lane of my row)
    bool lo;            // i < 8
    uint32_t m8, m89;   // all-ones in lane 8 / lanes 8 and 9 (where a carry out of limb 15 re-enters besides limb 0)
    uint32_t pb;        // limb i of p: 2^28 - 1, limb 8: 2^28 - 2
};
__device__ __forceinline__ Lane make_lane() {
    Lane L;
    const uint32_t l = threadIdx.x & 63u;
    L.i = l & 15u;
    L.row = l >> 4;
    L.i4 = (int)(L.i * 4);
    L.row4 = (int)((l & 48u) * 4);
    L.lo = L.i < 8;
    L.m8 = L.i == 8 ? ~0u : 0u;
    L.m89 = (L.i == 8 || L.i == 9) ? ~0u : 0u;
    L.pb = L.i == 8 ? M28 - 1 : M28;
    return L;
}

// dst[l] = src[(l - N) mod 16] inside every row
template <int N>
__device__ __forceinline__ uint32_t ror(uint32_t v) {
    static_assert(N >= 1 && N <= 15, "row_ror:1..15");
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x120 + N, 0xF, 0xF, false);   // every lane is written: no `old`
}
// low consumers (limbs 0..7) take vlo rotated, high consumers vhi rotated (DPP bank mask on the destination)
template <int N>
__device__ __forceinline__ uint32_t ror_split(const Lane &L, uint32_t vlo, uint32_t vhi) {
    if constexpr (N == 0) {
        return L.lo ? vlo : vhi;
    } else {
        const int t = __builtin_amdgcn_mov_dpp((int)vlo, 0x120 + N, 0xF, 0x3, false);   // lanes 8..15 are written next
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)vhi, 0x120 + N, 0xF, 0xC, false);
    }
}
// limb J of my row's element, in every lane of the row
template <int J>
__device__ __forceinline__ uint32_t bcast(const Lane &L, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(L.row4 + 4 * J, (int)v);
}
// the element of row K, in every row
template <int K>
__device__ __forceinline__ wfe from_row(const Lane &L, wfe v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(L.i4 + 64 * K, (int)v);
}
// rows permuted: my row receives the element of row src_row (per-lane value 0..3)
__device__ __forceinline__ wfe rows(const Lane &L, wfe v, uint32_t src_row) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_row * 64) + L.i4, (int)v);
}

// One parallel carry pass (fe_weak): every limb keeps 28 bits and receives its neighbour's excess;
// limb 15's excess re-enters at limbs 0 and 8.  Input limbs < 2^32; result < 2^28 + 2^5.
__device__ __forceinline__ wfe weak(const Lane &L, wfe r) {
    const uint32_t c = r >> 28;
    return (r & M28) + ror<1>(c) + (ror<9>(c) & L.m8);
}
// 64-bit column sums -> limbs (result < 2^28 + 8)
__device__ __forceinline__ wfe carry(const Lane &L, uint64_t s) {
    const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);
    const uint32_t t0 = lo & M28;
    const uint32_t t1 = (uint32_t)(s >> 28) & M28;
    const uint32_t t2 = hi >> 24;   // s >> 56
    // positions 16 and 17 fold onto (0, 8) and (1, 9): the rotation delivers the first of each pair
    const uint32_t r = t0 + ror<1>(t1) + ror<2>(t2) + (ror<9>(t1) & L.m8) + (ror<10>(t2) & L.m89);
    return weak(L, r);
}

__device__ __forceinline__ wfe add(wfe a, wfe b) { return a + b; }
template <int K>
__device__ __forceinline__ wfe sub(const Lane &L, wfe a, wfe b) { return a + (uint32_t)K * L.pb - b; }   // b <= K * p's limb

// Four products at once: row r gets a_r * b_r mod p.  mag(a) <= 7, mag(b) <= 5, mag(a) * mag(b) <= 6.7
// (gf28.hpp); result limbs < 2^28 + 8.
template <int J>
__device__ __forceinline__ void mul_step(const Lane &L, uint64_t &acc, const uint32_t (&a0)[8], const uint32_t (&a1)[8], wfe X,
                                         wfe Xp, wfe Y, wfe Yp) {
    const uint32_t u = ror_split<J>(L, X, Xp), v = ror_split<J>(L, Y, Yp);
    acc += (uint64_t)a0[J] * u;
    asm("" : "+v"(acc));   // keep the accumulation a chain of v_mad_u64_u32 (see gf28.hpp acc_t::mac)
    acc += (uint64_t)a1[J] * v;
    asm("" : "+v"(acc));
}
__device__ __forceinline__ wfe mul(const Lane &L, wfe a, wfe b) {
    // the multiplier's sixteen limbs, each broadcast inside its row: all sixteen crossbar requests go
    // out first, so their latency overlaps instead of being paid once per step
    uint32_t a0[8], a1[8];
    a0[0] = bcast<0>(L, a); a0[1] = bcast<1>(L, a); a0[2] = bcast<2>(L, a); a0[3] = bcast<3>(L, a);
    a0[4] = bcast<4>(L, a); a0[5] = bcast<5>(L, a); a0[6] = bcast<6>(L, a); a0[7] = bcast<7>(L, a);
    a1[0] = bcast<8>(L, a); a1[1] = bcast<9>(L, a); a1[2] = bcast<10>(L, a); a1[3] = bcast<11>(L, a);
    a1[4] = bcast<12>(L, a); a1[5] = bcast<13>(L, a); a1[6] = bcast<14>(L, a); a1[7] = bcast<15>(L, a);
    const uint32_t t1 = ror<8>(b);          // the other half's limb
    const uint32_t sb = b + t1;              // (b0 + b1) limb, in both halves
    const uint32_t X = b;                    // [b0 | b1]       x a0_j for low consumers
    const uint32_t Xp = L.lo ? sb : b;       // [b0+b1 | b1]    x a0_j for high consumers
    const uint32_t Y = L.lo ? t1 : sb;       // [b1 | b0+b1]    x a1_j for low consumers
    const uint32_t Yp = L.lo ? sb + t1 : sb; // [b0+2b1 | b0+b1] x a1_j for high consumers
    uint64_t acc = 0;
    mul_step<0>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<1>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<2>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<3>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<4>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<5>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<6>(L, acc, a0, a1, X, Xp, Y, Yp);
    mul_step<7>(L, acc, a0, a1, X, Xp, Y, Yp);
    return carry(L, acc);
}
// a * w, w < 2^18 (the curve constants); result < 2^28 + 2^20
__device__ __forceinline__ wfe mulw(const Lane &L, wfe a, uint32_t w) {
    const uint64_t p = (uint64_t)a * w;
    const uint32_t t1 = (uint32_t)(p >> 28);
    return ((uint32_t)p & M28) + ror<1>(t1) + (ror<9>(t1) & L.m8);
}

// ---------------------------------------------------------------- points: rows (X, Y, Z, T) of one register

__device__ __forceinline__ wfe identity(const Lane &L) {   // (0, 1, 1, 0)
    return (L.i == 0 && (L.row == 1 || L.row == 2)) ? 1u : 0u;
}

// P <- 2P (src/goldilocks.c:232-254; T always produced, it costs nothing here)
__device__ __forceinline__ wfe dbl(const Lane &L, wfe P) {
    const wfe x = from_row<0>(L, P), y = from_row<1>(L, P);
    const wfe v1 = L.row == 3 ? x + y : P;                  // (X, Y, Z, X+Y), mag <= 2
    const wfe q = mul(L, v1, v1);                           // (C, A, ZZ, SS)
    const wfe c = from_row<0>(L, q), a = from_row<1>(L, q), zz = from_row<2>(L, q), ss = from_row<3>(L, q);
    const wfe d = c + a;                                    // X^2 + Y^2              mag 2
    const wfe tt = sub<2>(L, a, c);                         // Y^2 - X^2              mag 3
    const wfe b = weak(L, sub<3>(L, ss, d));                // 2XY                    mag 1
    const wfe e = weak(L, sub<4>(L, zz + zz, tt));          // 2Z^2 - (Y^2 - X^2)     mag 1
    const wfe av = (L.row & 1u) ? d : e;                    // (E, D, E, D)
    const wfe bv = (L.row == 0 || L.row == 3) ? b : tt;     // (B, T', T', B)
    return mul(L, av, bv);                                  // (E*B, D*T', E*T', D*B) = (X, Y, Z, T)
}

// point -> projective niels entry, rows (a, b, cn, z) = (Y-X, Y+X, 2*39082*T, 2Z)   (src/goldilocks.c:280-288)
__device__ __forceinline__ wfe to_pniels(const Lane &L, wfe P, uint32_t swap_row) {
    const wfe p0 = rows(L, P, swap_row);                    // (Y, X, T, Z)
    const wfe ab = weak(L, L.row == 0 ? sub<2>(L, p0, P) : p0 + P);
    const wfe cn = mulw(L, p0, TWO_EFF_D);
    return L.row < 2 ? ab : (L.row == 2 ? cn : p0 + p0);
}

// P <- P +- entry, ev = rows (a, b, cn, z) ALREADY swapped in its first two rows when neg
// (src/goldilocks.c:314-380)
__device__ __forceinline__ wfe add_entry(const Lane &L, wfe P, wfe ev, bool neg, uint32_t swap_row) {
    const wfe p0 = rows(L, P, swap_row);                    // (Y, X, T, Z)
    const wfe av = L.row == 0 ? sub<2>(L, p0, P) : (L.row == 1 ? p0 + P : p0);   // (Y-X, X+Y, T, Z)  mag 3, 2, 1, 1
    const wfe r = mul(L, av, ev);                           // (A, B, Cn, ZZ)
    const wfe A = from_row<0>(L, r), B = from_row<1>(L, r), Cn = from_row<2>(L, r), ZZ = from_row<3>(L, r);
    const wfe E = weak(L, sub<2>(L, B, A));                 // mag 1
    const wfe H = A + B;                                    // mag 2
    const wfe zm = sub<2>(L, ZZ, Cn);                       // mag 3
    const wfe zp = ZZ + Cn;                                 // mag 2
    const wfe F = neg ? zm : zp, G = neg ? zp : zm;
    const wfe a2 = L.row == 3 ? E : (L.row == 1 ? G : F);   // (F, G, F, E)
    const wfe b2 = L.row == 0 ? E : (L.row == 2 ? G : H);   // (E, H, G, H)
    return mul(L, a2, b2);                                  // (F*E, G*H, F*G, E*H) = (X, Y, Z, T)
}

// ---------------------------------------------------------------- ABI I/O: 4 x 8 x u64 <-> rows

__device__ __forceinline__ wfe load_point(const Lane &L, const uint64_t *p) {
    const uint64_t l56 = p[8 * L.row + (L.i >> 1)];
    const uint32_t v = (L.i & 1u) ? (uint32_t)(l56 >> 28) : (uint32_t)l56 & M28;   // the odd limb keeps the excess above 2^56
    return weak(L, v);
}
__device__ __forceinline__ void store_point(const Lane &L, uint64_t *p, wfe P) {
    const wfe w = weak(L, P);
    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x101, 0xF, 0xF, false);   // row_shl:1: lane l gets lane l+1 (lane 15: 0, unused)
    if (!(L.i & 1u)) p[8 * L.row + (L.i >> 1)] = (uint64_t)w + ((uint64_t)nb << 28);
}

// ---------------------------------------------------------------- the ladder

constexpr int TABLE_WORDS = 17 * 64;   // 16 entries + the build step, 64 words each

struct WaveTable {
    uint32_t *t;   // this wave's table in LDS
    __device__ __forceinline__ void store(const Lane &L, int k, wfe ev) const { t[k * 64 + (threadIdx.x & 63u)] = ev; }
    __device__ __forceinline__ wfe load(const Lane &L, int k) const { return t[k * 64 + (threadIdx.x & 63u)]; }
    // entry idx with rows a / b exchanged when neg: every entry is read, one is kept.  The digit may be secret: a lane
    // reads its OWN row whatever the sign (no address depends on the digit, tools/isa_audit.py) and the exchange of rows
    // 0 and 1 is a register move between lanes (ds_bpermute_b32) kept by a select.
    __device__ __forceinline__ wfe lookup(const Lane &L, uint32_t idx, bool neg) const {
        const uint32_t *q = t + L.row * 16 + L.i;
        wfe r = q[0];
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const wfe v = q[k * 64];
            r = idx == (uint32_t)k ? v : r;
        }
        const wfe exchanged = rows(L, r, L.row < 2 ? (L.row ^ 1u) : L.row);
        return neg ? exchanged : r;
    }
};

// The 16 odd multiples of B as projective niels into the wave's LDS table (src/goldilocks.c:382-403).
__device__ __forceinline__ void build_table(const Lane &L, const WaveTable &tab, wfe B) {
    const uint32_t swap_row = L.row ^ 1u;    // rows (1, 0, 3, 2)
    tab.store(L, 16, to_pniels(L, dbl(L, B), swap_row));
    tab.store(L, 0, to_pniels(L, B, swap_row));
    wfe acc = B;
#pragma unroll 1
    for (int e = 1; e < 16; e++) {
        acc = add_entry(L, acc, tab.load(L, 16), false, swap_row);
        tab.store(L, e, to_pniels(L, acc, swap_row));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// scalar * (the table's point): 90 signed 5-bit windows over the table (index-independent lookups).
// bits: the wave's 15-word LDS slot.
__device__ __forceinline__ wfe walk_table(const Lane &L, const WaveTable &tab, uint32_t *bits, const sc &k) {
    const uint32_t swap_row = L.row ^ 1u;
    const sc r = sc_recode_signed(k);
#pragma unroll
    for (int w = 0; w < 14; w++) bits[w] = r.w[w];    // every lane writes the same words
    bits[14] = 0;
    struct Bits {
        const uint32_t *p;
        __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
    } rb{bits};
    uint32_t idx;
    bool neg;
    signed_digit(window5(rb, 445), idx, neg);
    wfe acc = add_entry(L, identity(L), tab.lookup(L, idx, neg), neg, swap_row);
#pragma unroll 1
    for (int pos = 440; pos >= 0; pos -= 5) {
        signed_digit(window5(rb, pos), idx, neg);
#pragma unroll 1
        for (int j = 0; j < 5; j++) acc = dbl(L, acc);
        acc = add_entry(L, acc, tab.lookup(L, idx, neg), neg, swap_row);
    }
    return acc;
}
// out = scalar * base for ONE operation handled by this wave.
__device__ __forceinline__ wfe scalarmul(const Lane &L, const WaveTable &tab, uint32_t *bits, wfe B, const sc &k) {
    build_table(L, tab, B);
    return walk_table(L, tab, bits, k);
}

// s1 * P1 + s2 * P2 with both tables built: ONE ladder of `windows` 5-bit windows, two additions per window
// (src/goldilocks.c:467-541).  b1 / b2: the recoded scalars' words; flip1 / flip2 negate a point by its digits.
template <class BITS>
__device__ __forceinline__ wfe walk_two_tables(const Lane &L, const WaveTable &tab1, const WaveTable &tab2, const BITS &b1,
                                               const BITS &b2, int windows, bool flip1, bool flip2) {
    const uint32_t swap_row = L.row ^ 1u;
    uint32_t idx;
    bool neg;
    const int top = 5 * (windows - 1);
    signed_digit(window5(b1, top), idx, neg);
    neg = neg != flip1;
    wfe V = add_entry(L, identity(L), tab1.lookup(L, idx, neg), neg, swap_row);
    signed_digit(window5(b2, top), idx, neg);
    neg = neg != flip2;
    V = add_entry(L, V, tab2.lookup(L, idx, neg), neg, swap_row);
#pragma unroll 1
    for (int pos = top - 5; pos >= 0; pos -= 5) {
#pragma unroll 1
        for (int j = 0; j < 5; j++) V = dbl(L, V);
        signed_digit(window5(b1, pos), idx, neg);
        neg = neg != flip1;
        V = add_entry(L, V, tab1.lookup(L, idx, neg), neg, swap_row);
        signed_digit(window5(b2, pos), idx, neg);
        neg = neg != flip2;
        V = add_entry(L, V, tab2.lookup(L, idx, neg), neg, swap_row);
    }
    return V;
}

// ---------------------------------------------------------------- canonical form, predicates (per row)

// dst[l] = src[l - 1], lane 0 of every row gets 0
__device__ __forceinline__ uint32_t shr1_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1, bound_ctrl: 0 shifted in
}
// Full carry/borrow propagation of signed limbs t_i (|t_i| < 2^30): returns limbs in [0, 2^28) and in
// `top` (every lane of the row) the signed carry out of limb 15: value = sum limb_i 2^(28 i) + top 2^448.
__device__ __forceinline__ wfe ripple(const Lane &L, int32_t t, int32_t &top) {
    int32_t out = 0;
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        const int32_t c = t >> 28;                   // arithmetic: borrows are negative carries
        out += c;                                    // meaningful in lane 15 only
        t = (t & (int32_t)M28) + (int32_t)shr1_zero((uint32_t)c);
    }
    top = (int32_t)bcast<15>(L, (uint32_t)out);
    return (uint32_t)t;
}
// Canonical representative in [0, p), limbs < 2^28  (cf. gf_strong_reduce, src/f_generic.c:71-105).  Input mag <= 15.
__device__ __forceinline__ wfe strong(const Lane &L, wfe a) {
    const wfe w = weak(L, weak(L, a));               // value < 2p, limbs <= 2^28
    int32_t top;
    const wfe t = ripple(L, (int32_t)w - (int32_t)L.pb, top);      // w - p
    // top == -1: w < p, add p back (the carry out of that addition cancels the borrow)
    int32_t dummy;
    return ripple(L, (int32_t)t + (int32_t)(L.pb & (uint32_t)top), dummy);
}
// a == 0 mod p, per row (the answer is the same in every lane of a row)
__device__ __forceinline__ bool is_zero(const Lane &L, wfe a) {
    uint32_t x = strong(L, a);
    x |= ror<8>(x);
    x |= ror<4>(x);
    x |= ror<2>(x);
    x |= ror<1>(x);
    return x == 0;
}
__device__ __forceinline__ bool eq(const Lane &L, wfe a, wfe b) { return is_zero(L, sub<2>(L, a, weak(L, b))); }   // a, b mag <= 2
__device__ __forceinline__ bool lobit(const Lane &L, wfe a) { return bcast<0>(L, strong(L, a)) & 1u; }
__device__ __forceinline__ wfe neg(const Lane &L, wfe a) { return sub<2>(L, 0u, weak(L, a)); }                    // mag 2
__device__ __forceinline__ wfe one(const Lane &L) { return L.i == 0 ? 1u : 0u; }
__device__ __forceinline__ wfe sqrn(const Lane &L, wfe x, int n) {
#pragma unroll 1
    for (int k = 0; k < n; k++) x = mul(L, x, x);
    return x;
}
// x^((p-3)/4) in every row at once; ok = (result^2 * x == 1) per row  (cf. gf_isr, src/f_arithmetic.c:14-46)
__device__ __forceinline__ wfe isr(const Lane &L, wfe x, bool &ok) {
    const wfe e1 = weak(L, x);
    const wfe e2 = mul(L, sqrn(L, e1, 1), e1);
    const wfe e3 = mul(L, sqrn(L, e2, 1), e1);
    const wfe e6 = mul(L, sqrn(L, e3, 3), e3);
    const wfe e9 = mul(L, sqrn(L, e6, 3), e3);
    const wfe e18 = mul(L, sqrn(L, e9, 9), e9);
    const wfe e19 = mul(L, sqrn(L, e18, 1), e1);
    const wfe e37 = mul(L, sqrn(L, e19, 18), e18);
    const wfe e74 = mul(L, sqrn(L, e37, 37), e37);
    const wfe e111 = mul(L, sqrn(L, e74, 37), e37);
    const wfe e222 = mul(L, sqrn(L, e111, 111), e111);
    const wfe e223 = mul(L, sqrn(L, e222, 1), e1);
    const wfe r = mul(L, sqrn(L, e223, 223), e222);
    ok = eq(L, mul(L, mul(L, r, r), e1), one(L));
    return r;
}

// 56 little-endian bytes -> limbs (lane i takes bits [28 i, 28 i + 28)); *below_p = the value is < p
// (cf. gf_deserialize, src/f_generic.c:49-68).  Rows read from their own string: p = this ROW's bytes.
__device__ __forceinline__ wfe deserialize(const Lane &L, const uint8_t *p, bool &below_p) {
    const uint32_t bit = 28 * L.i, b = bit >> 3, sh = bit & 7;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) {
        const uint32_t at = b + k;
        v |= (uint64_t)(at < 56 ? p[at] : 0) << (8 * k);
    }
    const wfe x = (uint32_t)(v >> sh) & M28;
    int32_t top;
    (void)ripple(L, (int32_t)x - (int32_t)L.pb, top);
    below_p = top != 0;                               // x - p went negative
    return x;
}

// RFC 8032 decoding followed by the 4-isogeny (cf. pt_decode_eddsa_words, src/goldilocks.c:949-1004), one
// encoding per ROW (enc = this row's 57 bytes): returns the row's (X, Y, Z, T) in four registers.
__device__ __forceinline__ bool decode_eddsa_rows(const Lane &L, const uint8_t *enc, wfe &X, wfe &Y, wfe &Z, wfe &T) {
    const uint32_t last = enc[56];
    const bool low = (last & 0x80u) != 0;
    bool ok = (last & 0x7fu) == 0, below, sq;
    const wfe y = deserialize(L, enc, below);
    ok = ok && below;
    const wfe y2 = mul(L, y, y);
    const wfe num = weak(L, sub<2>(L, one(L), y2));                            // 1 - y^2
    const wfe den = weak(L, one(L) + mulw(L, y2, NEG_EDWARDS_D));             // 1 - d y^2, d = -39081
    const wfe r = isr(L, mul(L, num, den), sq);
    ok = ok && sq;
    wfe x = mul(L, r, num);
    {   // cross-lane operations must run with every lane active: both candidates first, then a plain select
        const wfe nx = neg(L, x);
        const bool flip = lobit(L, x) != low;
        x = weak(L, flip ? nx : x);
    }
    // isogeny: like doubling with Z = 1 but E = 2 - D
    const wfe c = mul(L, x, x);
    const wfe d = c + y2;                                                       // mag 2
    const wfe xy = x + y;
    const wfe b = weak(L, sub<4>(L, mul(L, xy, xy), d));
    const wfe tt = weak(L, sub<2>(L, y2, c));
    const wfe e = weak(L, sub<4>(L, (L.i == 0 ? 2u : 0u), d));                  // 2 - D
    X = mul(L, e, b);
    Z = mul(L, tt, e);
    Y = mul(L, d, tt);
    T = mul(L, d, b);
    return ok;
}
// four registers holding row K's (X, Y, Z, T) -> one register with rows (X, Y, Z, T)
template <int K>
__device__ __forceinline__ wfe pack_point(const Lane &L, wfe X, wfe Y, wfe Z, wfe T) {
    const wfe x = from_row<K>(L, X), y = from_row<K>(L, Y), z = from_row<K>(L, Z), t = from_row<K>(L, T);
    return L.row == 0 ? x : (L.row == 1 ? y : (L.row == 2 ? z : t));
}

// acc += s*B through the base point's window table (affine niels, 12 uint4 per entry: scalarmul.hpp
// ladder_bwt; the entries are (Y-X, Y+X, 2*39082*T) / 2Z, i.e. projective niels with z = 1): rows
// (a, b, cn) read their limb, row 3 is the constant 1.  Public scalars only (verification).
template <class BITS>
__device__ __forceinline__ wfe add_base_multiple(const Lane &L, wfe acc, const BITS &bits, const uint4 *bwt) {
    const uint32_t swap_row = L.row ^ 1u;
    const BwtGeom g = GlobalBwt{bwt}.geom();
    const uint32_t *tab = reinterpret_cast<const uint32_t *>(bwt + BWT_HEADER_U4);
#pragma unroll 1
    for (uint32_t w = g.windows; w-- > 0;) {
        uint32_t idx;
        bool neg;
        signed_digit_bwt(window_bwt(bits, w, g.bits), g.bits, idx, neg);
        const uint32_t frow = (neg && L.row < 2) ? (L.row ^ 1u) : L.row;
        const uint32_t *e = tab + 48 * (((size_t)w << (g.bits - 1)) + idx);
        const wfe ev = L.row == 3 ? (L.i == 0 ? 1u : 0u) : e[frow * 16 + L.i];
        acc = add_entry(L, acc, ev, neg, swap_row);
    }
    return acc;
}

// One Ed448 verification by this wave with half-size scalars (eddsa.hpp ed448_verify_lattice, lattice.hpp):
// the two point decodings run in rows 0 and 1 of the same instruction stream; the hash and the short pair
// (rho, tau) of the challenge are computed by every lane alike; then ONE ladder of 45 windows over the tables of
// A and R (both in LDS, index-independent lookups), the base point's 28 additions, and V == identity.
// The signs of the lattice method ride on the digits: rho * (-+A) and |tau| * (-R) flip the digits' signs
// instead of negating the points.  bits: 2 x 16 words of the wave.
template <class STAGE>
__device__ __forceinline__ bool verify(const Lane &L, const WaveTable &tab_a, const WaveTable &tab_r, uint32_t *bits,
                                       const Ed448Msg &m, STAGE &stage, const uint4 *bwt) {
    const uint32_t swap_row = L.row ^ 1u;
    // row 0 decodes the public key, row 1 R (the other rows run along on the key)
    const uint8_t *enc = L.row == 1 ? m.a : m.b;
    wfe X, Y, Z, T;
    const bool okrow = decode_eddsa_rows(L, enc, X, Y, Z, T);
    const uint64_t okmask = __builtin_amdgcn_ballot_w64(okrow);
    const bool ok = (okmask & 1u) && ((okmask >> 16) & 1u);
    build_table(L, tab_a, pack_point<0>(L, X, Y, Z, T));
    build_table(L, tab_r, pack_point<1>(L, X, Y, Z, T));

    const LatticePair pr = ed448_verify_lattice_pair(m, stage);      // the same in every lane
    const bool flip_a = pr.tau_pos;                                  // PA = -A for a positive tau; PR = -R always
#pragma unroll
    for (int k = 0; k < 15; k++) {
        bits[k] = pr.b1[k];
        bits[16 + k] = pr.b2[k];
    }
    struct Bits {
        const uint32_t *p;
        __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
    } b1{bits}, b2{bits + 16};
    wfe V = walk_two_tables(L, tab_a, tab_r, b1, b2, LATTICE_WINDOWS, flip_a, true);
    // an even rho or |tau| was walked as the next odd number: one copy of its point too many (wave-uniform)
    if (pr.rho_even) V = add_entry(L, V, tab_a.lookup(L, 0, !flip_a), !flip_a, swap_row);   // - PA
    if (pr.tau_even) V = add_entry(L, V, tab_r.lookup(L, 0, false), false, swap_row);       // - PR = + R
    const sc rs = sc_recode_bwt(pr.ts, GlobalBwt{bwt});
#pragma unroll
    for (int k = 0; k < 14; k++) bits[k] = rs.w[k];
    bits[14] = 0;
    V = add_base_multiple(L, V, b1, bwt);                                  // + (|tau| S)*B
    const uint64_t zmask = __builtin_amdgcn_ballot_w64(is_zero(L, V));     // row 0 is X
    return ok && (zmask & 1u);
}

// ---------------------------------------------------------------- X448 (RFC 7748), one ladder per wave

// 56 canonical bytes of the element held by row K, written by that row's even lanes (7 bytes per limb pair)
template <int K>
__device__ __forceinline__ void store_bytes(const Lane &L, uint8_t *out, wfe canonical) {
    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)canonical, 0x101, 0xF, 0xF, false);   // lane l+1's limb
    const uint64_t w56 = (uint64_t)canonical | ((uint64_t)nb << 28);
    if (L.row == K && !(L.i & 1u)) {
        uint8_t *q = out + 7 * (L.i >> 1);
#pragma unroll
        for (int k = 0; k < 7; k++) q[k] = (uint8_t)(w56 >> (8 * k));
    }
}
// 1/x in every row (0 -> 0)   (cf. gf_invert, src/goldilocks.c:69-80)
__device__ __forceinline__ wfe invert(const Lane &L, wfe x) {
    bool ok;
    const wfe xr = weak(L, x);
    const wfe t = isr(L, mul(L, xr, xr), ok);
    return mul(L, mul(L, t, t), xr);
}

// The Montgomery ladder of src/goldilocks.c:1006-1076 with the state (x2, z2, x3, z3) in the four rows
// of one register: a step is three vector multiplications -- (A^2, B^2, D*A, C*B), then
// (AA*BB, E*(AA + a24 E), (DA+CB)^2, (DA-CB)^2), then the x1 factor of z3 -- and two row exchanges.
// Signs: rows 1 and 3 carry -B and -CB (the differences taken the other way round), which squares and
// the paired products absorb.  scalar: 14 words as given (clamped here).  Returns false iff the result is 0.
__device__ __forceinline__ bool x448(const Lane &L, uint8_t *out56, const uint8_t *base56, const uint32_t (&k)[14]) {
    bool below;
    const wfe x1 = deserialize(L, base56, below);        // the reference ignores the range check too (:1014)
    const wfe x1row = L.row == 3 ? x1 : one(L);           // third product: (1, 1, 1, x1)
    wfe S = L.row == 0 ? one(L) : (L.row == 1 ? 0u : (L.row == 2 ? x1 : one(L)));   // (1, 0, x1, 1)
    bool swap = false;
#pragma unroll 1
    for (int t = 447; t >= 0; t--) {
        uint32_t w = 0;                                   // word t >> 5 of the scalar: a uniform select chain keeps k in registers
#pragma unroll
        for (int j = 0; j < 14; j++) w = (t >> 5) == j ? k[j] : w;
        uint32_t bit = (w >> (t & 31)) & 1u;
        if (t < 2) bit = 0;                               // scalar[0] &= -COFACTOR
        if (t == 447) bit = 1;                            // top bit forced
        const bool kt = bit != 0, sw = swap != kt;
        swap = kt;
        const wfe other = rows(L, S, L.row ^ 2u);         // (x3, z3, x2, z2)
        S = sw ? other : S;
        const wfe partner = rows(L, S, L.row ^ 1u);       // (z2, x2, z3, x3)
        const wfe sum = S + partner;                      // (A, A, C, C)                mag 2
        const wfe diff = sub<2>(L, S, partner);           // (B, -B, D, -D)              mag 3
        const wfe a1 = (L.row == 0 || L.row == 3) ? sum : diff;          // (A, -B, D, C)
        const wfe kk = (L.row & 1u) ? weak(L, diff) : sum;               // rows 0, 1: (A, -B)
        const wfe b1 = rows(L, kk, L.row & 1u);                          // (A, -B, A, -B)
        const wfe r1 = mul(L, a1, b1);                                   // (AA, BB, DA, -CB)
        const wfe p1 = rows(L, r1, L.row ^ 1u);                          // (BB, AA, -CB, DA)
        const wfe s2 = r1 + p1;                                          // (., ., DA-CB, DA-CB)       mag 2
        const wfe d2 = weak(L, sub<2>(L, r1, p1));                       // (E, -E, DA+CB, -(DA+CB))   mag 1
        const wfe f = sub<2>(L, mulw(L, d2, 39081), p1);                 // row 1: a24*(-E) - AA       mag 3
        const wfe a2 = L.row == 0 ? r1 : (L.row == 3 ? s2 : d2);         // (AA, -E, DA+CB, DA-CB)
        const wfe b2 = L.row == 0 ? p1 : (L.row == 1 ? f : (L.row == 2 ? d2 : s2));   // (BB, -(AA+a24 E), DA+CB, DA-CB)
        const wfe r2 = mul(L, a2, b2);                                   // (x2', z2', x3', (DA-CB)^2)
        S = mul(L, r2, x1row);                                           // z3' = x1 (DA-CB)^2
    }
    const wfe fin = rows(L, S, L.row ^ 2u);
    S = swap ? fin : S;                                   // rows 0, 1: the result x2 / z2
    const wfe zi = invert(L, from_row<1>(L, S));
    const wfe r = strong(L, mul(L, from_row<0>(L, S), zi));
    store_bytes<0>(L, out56, r);
    uint32_t nz = r;
    nz |= ror<8>(nz); nz |= ror<4>(nz); nz |= ror<2>(nz); nz |= ror<1>(nz);
    return __builtin_amdgcn_readfirstlane(nz) != 0;       // row 0
}

// ---------------------------------------------------------------- fixed base: the 5x5x18 comb, key derivation, signing

// scalar * G through a comb table in this library's niels form (80 entries x 48 words, global memory, as
// k_import_comb leaves it): the loop of ladder_comb (src/goldilocks.c:830-877); every lookup reads the 16
// entries of its comb (one word per lane each) and keeps one -- index-independent, whatever the table mode.
template <class BITS>
__device__ __forceinline__ wfe comb_scalarmul(const Lane &L, const uint4 *comb, const BITS &bits) {
    const uint32_t swap_row = L.row ^ 1u;
    const uint32_t *tab = reinterpret_cast<const uint32_t *>(comb);
    wfe acc = identity(L);
#pragma unroll 1
    for (int i = 17; i >= 0; i--) {
        if (i != 17) acc = dbl(L, acc);
#pragma unroll 1
        for (int j = 0; j < 5; j++) {
            uint32_t idx;
            bool neg;
            signed_digit(comb_teeth(bits, i, j), idx, neg);
            // (a lane reads its own row whatever the digit's sign; rows a / b are exchanged between lanes afterwards)
            const uint32_t *q = tab + 48 * 16 * j + (L.row < 3 ? L.row : 0u) * 16 + L.i;
            wfe r = q[0];
#pragma unroll
            for (int k = 1; k < 16; k++) {
                const wfe v = q[48 * k];
                r = idx == (uint32_t)k ? v : r;
            }
            const wfe exchanged = rows(L, r, L.row < 2 ? (L.row ^ 1u) : L.row);
            r = neg ? exchanged : r;
            const wfe ev = L.row == 3 ? (L.i == 0 ? 1u : 0u) : r;       // affine entries: z = 1
            acc = add_entry(L, acc, ev, neg, swap_row);
        }
    }
    return acc;
}

// RFC 8032 encoding of 4*P (cf. pt_encode_eddsa_words, src/goldilocks.c:905-946): 57 bytes at out (any
// address space).  The squares of the dual isogeny are the first half of a doubling.
__device__ __forceinline__ void encode_eddsa(const Lane &L, uint8_t *out57, wfe P) {
    const wfe x = from_row<0>(L, P), y = from_row<1>(L, P);
    const wfe v1 = L.row == 3 ? x + y : P;
    const wfe q = mul(L, v1, v1);                           // (X^2, Y^2, Z^2, (X+Y)^2)
    const wfe c = from_row<0>(L, q), a = from_row<1>(L, q), zz = from_row<2>(L, q), ss = from_row<3>(L, q);
    const wfe u = c + a;                                    // mag 2
    const wfe yy = weak(L, sub<4>(L, ss, u));               // 2XY
    const wfe z = weak(L, sub<2>(L, a, c));                 // Y^2 - X^2
    const wfe tt = weak(L, sub<2>(L, zz + zz, z));          // 2Z^2 - (Y^2 - X^2)
    const wfe av = L.row == 0 ? tt : u;                     // (tt, u, u, .)
    const wfe bv = L.row == 0 ? yy : (L.row == 1 ? z : tt); // (yy, z, tt, .)
    const wfe n = mul(L, av, bv);                           // (xn, yn, zn, .)
    const wfe zi = invert(L, from_row<2>(L, n));
    const wfe aff = mul(L, n, zi);                          // (x, y, 1, .) affine
    const wfe can = strong(L, aff);
    store_bytes<1>(L, out57, can);
    const uint32_t sign = __builtin_amdgcn_readfirstlane(can) & 1u;   // row 0, limb 0: lobit(x)
    if ((threadIdx.x & 63u) == 0) out57[56] = (uint8_t)(sign << 7);
}
// (y/x)^2 serialized (cf. x448_public_finish, src/goldilocks.c:1102-1113)
__device__ __forceinline__ void encode_x448(const Lane &L, uint8_t *out56, wfe P) {
    const wfe xi = invert(L, from_row<0>(L, P));
    const wfe r = mul(L, from_row<1>(L, P), xi);
    store_bytes<0>(L, out56, strong(L, mul(L, r, r)));
}

// ---------------------------------------------------------------- Elligator 2 hash-to-curve

// One 56-byte string per ROW -> that row's point (X, Y, Z, T in four registers)   (cf. pt_from_hash_words,
// src/elligator.c:32-83).  The rows may hash different strings: the uniform variant maps its two halves at once.
__device__ __forceinline__ void from_hash_rows(const Lane &L, const uint8_t *str56, wfe &X, wfe &Y, wfe &Z, wfe &T) {
    bool below, square;
    const wfe r0 = strong(L, deserialize(L, str56, below));          // any 448-bit string, reduced mod p
    const wfe r = weak(L, neg(L, mul(L, r0, r0)));                    // r = -r0^2 (qnr = -1)
    const wfe rm1 = weak(L, sub<2>(L, r, one(L)));                    // r - 1
    const wfe drd = weak(L, neg(L, mulw(L, rm1, NEG_EDWARDS_D)));     // d r - d, d = -39081
    const wfe a = drd + one(L);                                       // mag 2
    const wfe b = weak(L, sub<2>(L, drd, r));
    const wfe D = mul(L, a, b);                                       // (dr - d + 1)(dr - d - r)
    const wfe N = weak(L, mulw(L, r + one(L), 78163));                // (r + 1)(1 - 2d)
    const wfe i = isr(L, mul(L, D, N), square);
    const wfe e = mul(L, i, square ? one(L) : r0);
    wfe s = mul(L, N, e);
    {   // cross-lane operations with every lane active: both candidates, then a plain select
        const wfe ns = neg(L, s);
        const bool flip = lobit(L, s) != !square;                     // negate iff lobit(s) ^ ~square
        s = weak(L, flip ? ns : s);
    }
    const wfe c = weak(L, mulw(L, e, 78163));
    wfe t = mul(L, mul(L, mul(L, c, c), rm1), N);
    {
        const wfe nt = neg(L, t);
        t = weak(L, square ? nt : t);
    }
    t = weak(L, sub<2>(L, t, one(L)));
    const wfe s2 = mul(L, s, s);
    const wfe two_s = s + s;                                          // mag 2
    const wfe ep = s2 + one(L);                                       // 1 + s^2, mag 2
    const wfe em = weak(L, sub<2>(L, one(L), s2));                    // 1 - s^2
    T = mul(L, two_s, ep);
    X = mul(L, two_s, t);
    Y = mul(L, ep, em);
    Z = mul(L, em, t);
}

// ---------------------------------------------------------------- precompute: the 5 x 5 x 18 comb of one point

// goldilocks_448_precompute by ONE wave (cf. k_precompute, src/goldilocks.c:755-818): the chain of 449 doublings
// through the teeth 2^(18a) P, a Gray-code walk over the 16 sign patterns of each comb (one addition of a
// doubled tooth per entry), then Montgomery's trick over the 80 values 2Z and the reference's table format
// (80 x {a, b, c}, canonical 56-bit limbs).  lds: 80 entries + 4 doubled teeth of 64 words, 80 prefixes of 16.
constexpr int PRECOMP_WAVE_WORDS = 84 * 64 + 80 * 16;
__device__ __forceinline__ void precompute(const Lane &L, uint32_t *lds, uint64_t *table, wfe P) {
    const uint32_t swap_row = L.row ^ 1u, me = threadIdx.x & 63u;
    uint32_t *entry = lds, *teeth = lds + 80 * 64, *prefix = lds + 84 * 64;
    wfe working = P;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        wfe start = working;   // becomes tooth_0 + ... + tooth_4 of this comb
#pragma unroll 1
        for (int k = 0; k < 5; k++) {
            if (k) start = add_entry(L, start, to_pniels(L, working, swap_row), false, swap_row);
            if (k == 4 && j == 4) break;
            working = dbl(L, working);
            if (k < 4) teeth[k * 64 + me] = to_pniels(L, working, swap_row);   // 2 * tooth_k
#pragma unroll 1
            for (int d = 0; d < 17; d++) working = dbl(L, working);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (uint32_t g = 0;; g++) {
            const uint32_t gray = g ^ (g >> 1);
            const uint32_t idx = (((j + 1) << 4) - 1) ^ gray;
            entry[idx * 64 + me] = to_pniels(L, start, swap_row);              // rows (Y-X, Y+X, 2*39082*T, 2Z)
            if (g >= 15) break;
            const uint32_t delta = (g + 1) ^ ((g + 1) >> 1) ^ gray;            // the Gray bit that flips
            const uint32_t k = 31 - __clz(delta);
            const bool neg = (gray & (1u << k)) == 0;
            const uint32_t frow = (neg && L.row < 2) ? (L.row ^ 1u) : L.row;   // a / b exchanged for a subtraction
            start = add_entry(L, start, teeth[k * 64 + frow * 16 + L.i], neg, swap_row);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Montgomery's trick over the 80 values 2Z (src/goldilocks.c:703-726); every row carries the same element
    wfe acc = one(L);
#pragma unroll 1
    for (int e = 0; e < 80; e++) {
        if (L.row == 0) prefix[e * 16 + L.i] = acc;
        acc = mul(L, acc, entry[e * 64 + 48 + L.i]);                           // row 3 of the entry: 2Z
    }
    wfe inv = invert(L, acc);
#pragma unroll 1
    for (int e = 79; e >= 0; e--) {
        const wfe zi = mul(L, inv, prefix[e * 16 + L.i]);
        inv = mul(L, inv, entry[e * 64 + 48 + L.i]);
        const wfe ev = entry[e * 64 + me];
        wfe abc = mul(L, ev, zi);                                              // ((Y-X)/2Z, (Y+X)/2Z, 78164 T / 2Z, .)
        {   // c = 2 d' T / 2Z = -(78164 T / 2Z)
            const wfe n = neg(L, abc);
            abc = L.row == 2 ? n : abc;
        }
        const wfe can = strong(L, abc);
        const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)can, 0x101, 0xF, 0xF, false);   // lane l+1's limb
        if (L.row < 3 && !(L.i & 1u)) table[24 * e + 8 * L.row + (L.i >> 1)] = (uint64_t)can | ((uint64_t)nb << 28);
    }
}

// ---------------------------------------------------------------- decaf wire format (direct_scalarmul)

__device__ __forceinline__ wfe factor(const Lane &L) { return FACTOR28[L.i]; }   // 1/sqrt(39082/39081 - 1), point.hpp

// Decaf decoding (cf. pt_decode_words, src/goldilocks.c:142-176): every row computes the same field
// elements; P = rows (X, Y, 1, T).  Returns success (wave-uniform).
__device__ __forceinline__ bool decode(const Lane &L, const uint8_t *ser56, bool allow_identity, wfe &P) {
    bool below, sq;
    const wfe s = deserialize(L, ser56, below);
    const bool zero = is_zero(L, s), odd = lobit(L, s);
    bool ok = below && (allow_identity || !zero) && !odd;
    const wfe s2 = mul(L, s, s);
    const wfe den = weak(L, sub<2>(L, one(L), s2));                  // 1 - s^2
    const wfe ynum = one(L) + s2;                                    // 1 + s^2       mag 2
    const wfe den2 = mul(L, den, den);
    const wfe num = weak(L, den2 + mulw(L, s2, FOUR_EFF_D));         // den^2 - 4 d' s^2
    const wfe r = isr(L, mul(L, num, den2), sq);
    ok = ok && sq;
    const wfe tmp = mul(L, r, den);
    const wfe y = mul(L, tmp, ynum);
    wfe w = mul(L, tmp, s);
    w = w + w;                                                       // 2 s isr den   mag 2
    wfe x = mul(L, mul(L, w, r), num);
    {   // cross-lane operations with every lane active: both candidates, then a plain select
        const wfe nx = neg(L, x);
        const bool flip = lobit(L, mul(L, w, factor(L)));
        x = weak(L, flip ? nx : x);
    }
    const wfe t = mul(L, x, y);
    P = L.row == 0 ? x : (L.row == 1 ? y : (L.row == 2 ? one(L) : t));
    return ok;
}
// Decaf encoding (cf. pt_encode_words, src/goldilocks.c:98-140): 56 canonical bytes at out
__device__ __forceinline__ void encode(const Lane &L, uint8_t *out56, wfe P) {
    const wfe x = from_row<0>(L, P), z = from_row<2>(L, P), t = from_row<3>(L, P);
    const wfe num = mul(L, x + t, weak(L, sub<2>(L, x, t)));         // (X+T)(X-T)
    const wfe x2 = mul(L, x, x);
    const wfe t2 = mulw(L, mul(L, x2, num), NEG_EDWARDS_D);          // 39081 X^2 num
    bool sq;
    const wfe r = isr(L, t2, sq);
    wfe ratio = mul(L, r, num);
    {
        const wfe nr = neg(L, ratio);
        const bool negx = lobit(L, mul(L, ratio, factor(L)));
        ratio = weak(L, negx ? nr : ratio);
    }
    const wfe t3 = weak(L, sub<2>(L, mul(L, ratio, z), t));
    const wfe t4 = mulw(L, mul(L, t3, x), NEG_EDWARDS_D);
    wfe sres = mul(L, t4, r);
    {
        const wfe ns = neg(L, sres);
        const bool lo = lobit(L, sres);
        sres = lo ? ns : sres;
    }
    store_bytes<0>(L, out56, strong(L, sres));
}

struct WaveBits {   // the wave's recoded scalar in LDS
    const uint32_t *p;
    __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
};
__device__ __forceinline__ WaveBits put_bits(uint32_t *bits, const sc &r) {
#pragma unroll
    for (int k = 0; k < 14; k++) bits[k] = r.w[k];
    bits[14] = 0;
    return WaveBits{bits};
}

// pk = encode((clamp(SHAKE256(sk)[0:57]) / 4) * B)   (cf. ed448_derive_core, src/eddsa.c:98-147)
template <class STAGE>
__device__ __forceinline__ void derive(const Lane &L, uint8_t *pk57, const uint8_t *sk57, const uint4 *comb, uint32_t *bits,
                                       STAGE &stage) {
    Ed448Msg m;
    m.a = sk57; m.alen = 57; m.b = sk57; m.blen = 0; m.msg = sk57; m.msglen = 0;
    m.ctx = sk57; m.ctxlen = 0; m.ph = 0; m.dom = false;
    uint32_t w[29];
    shake256_114(w, m, 57, stage);
    ed448_clamp_words(w);
    const sc secret = sc_halve(sc_halve(sc_decode_long_words<57>(w)));
    encode_eddsa(L, pk57, comb_scalarmul(L, comb, put_bits(bits, sc_recode_signed(secret))));
}

// RFC 8032 signature (cf. ed448_sign_core, src/eddsa.c:149-230).  lds57: two 64-byte LDS buffers of this
// wave, for the hashed-key seed and for R (both are hashed again).
template <class STAGE>
__device__ __forceinline__ void sign(const Lane &L, uint8_t *sig114, const uint8_t *sk57, const uint8_t *pk57, const uint8_t *msg,
                                     uint32_t msglen, uint32_t ph, const uint8_t *ctx, uint32_t ctxlen, const uint4 *comb,
                                     uint32_t *bits, uint8_t *lds128, STAGE &stage) {
    uint8_t *seed = lds128, *rbytes = lds128 + 64;
    Ed448Msg m;
    m.a = sk57; m.alen = 57; m.b = sk57; m.blen = 0; m.msg = sk57; m.msglen = 0;
    m.ctx = ctx; m.ctxlen = 0; m.ph = 0; m.dom = false;
    uint32_t w[29];
    shake256_114(w, m, 57, stage);                             // expanded = secret(57) | seed(57)
    for (int i = 0; i < 57; i++) seed[i] = (uint8_t)(w[(57 + i) >> 2] >> (8 * ((57 + i) & 3)));   // every lane, same bytes
    uint32_t sw[15];
#pragma unroll
    for (int i = 0; i < 15; i++) sw[i] = w[i];
    ed448_clamp_words(sw);
    const sc secret = sc_decode_long_words<57>(sw);
    m.a = seed; m.alen = 57; m.blen = 0; m.msg = msg; m.msglen = msglen;
    m.ctx = ctx; m.ctxlen = ctxlen; m.ph = ph ? 1u : 0u; m.dom = true;
    shake256_114(w, m, m.total(), stage);
    const sc nonce = sc_decode_long_words<114>(w);
    const wfe R = comb_scalarmul(L, comb, put_bits(bits, sc_recode_signed(sc_halve(sc_halve(nonce)))));
    encode_eddsa(L, rbytes, R);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const Ed448Msg c = ed448_challenge_string(rbytes, pk57, msg, msglen, ph, ctx, ctxlen);
    shake256_114(w, c, c.total(), stage);
    const sc challenge = sc_decode_long_words<114>(w);
    const sc resp = sc_add(sc_mul(challenge, secret), nonce);
    const uint32_t l = threadIdx.x & 63u;
    if (l < 57) sig114[l] = rbytes[l];
    if (l == 56) sig114[113] = 0;
    // S: 56 bytes of resp + a zero byte; lane l writes byte l (the words are the same in every lane)
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) word = (l >> 2) == (uint32_t)j ? resp.w[j] : word;
    if (l < 56) sig114[57 + l] = (uint8_t)(word >> (8 * (l & 3)));
    // the seed does not stay behind
    if (l < 16) reinterpret_cast<uint32_t *>(seed)[l] = 0;
}

}  // namespace wc
}  // namespace gd
