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
    // entry idx with rows a / b exchanged when neg: every entry is read, one is kept
    __device__ __forceinline__ wfe lookup(const Lane &L, uint32_t idx, bool neg) const {
        const uint32_t frow = (neg && L.row < 2) ? (L.row ^ 1u) : L.row;
        const uint32_t *q = t + frow * 16 + L.i;
        wfe r = q[0];
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const wfe v = q[k * 64];
            r = idx == (uint32_t)k ? v : r;
        }
        return r;
    }
};

// out = scalar * base for ONE operation handled by this wave.  bits: the wave's 15-word LDS slot.
__device__ __forceinline__ wfe scalarmul(const Lane &L, const WaveTable &tab, uint32_t *bits, wfe B, const sc &k) {
    const uint32_t swap_row = L.row ^ 1u;    // rows (1, 0, 3, 2)
    const sc r = sc_recode_signed(k);
#pragma unroll
    for (int w = 0; w < 14; w++) bits[w] = r.w[w];    // every lane writes the same words
    bits[14] = 0;
    // table of odd multiples (src/goldilocks.c:382-403)
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
    struct Bits {
        const uint32_t *p;
        __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
    } rb{bits};
    uint32_t idx;
    bool neg;
    signed_digit(window5(rb, 445), idx, neg);
    acc = add_entry(L, identity(L), tab.lookup(L, idx, neg), neg, swap_row);
#pragma unroll 1
    for (int pos = 440; pos >= 0; pos -= 5) {
        signed_digit(window5(rb, pos), idx, neg);
#pragma unroll 1
        for (int j = 0; j < 5; j++) acc = dbl(L, acc);
        acc = add_entry(L, acc, tab.lookup(L, idx, neg), neg, swap_row);
    }
    return acc;
}

}  // namespace wc
}  // namespace gd
