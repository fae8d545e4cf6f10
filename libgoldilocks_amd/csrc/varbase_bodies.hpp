// varbase_bodies.hpp -- bodies of the kernels that multiply a caller's point by a scalar, in two families:
//   * digit-addressed window tables (5-bit signed windows, the digit picks the address of its entry in the
//     lane's table): public scalars -- verification, base_double_scalarmul_non_secret, or a caller who asked for
//     GOLDILOCKS_AMD_TABLES_FAST in goldilocks_448_point_double_scalarmul.  Instantiated by kernels_verify_lanes.hip.
//   * NO table at all (montgomery.hpp: a Montgomery ladder of selects): the library's default for every entry
//     point whose scalar may be secret -- the counterpart of the reference's constant_time_lookup
//     (src/include/constant_time.h:134-183), which goldilocks_448_point_scalarmul / _double_scalarmul /
//     _dual_scalarmul / direct_scalarmul all use (src/goldilocks.c:437-442, :500-520, :590-610).
//     Instantiated by kernels_varbase_ct.hip.
#pragma once
#include "kernels.hpp"
#include "fixed_bodies.hpp"
#include "montgomery2d.hpp"
#include "montgomery.hpp"

namespace gd {

// table `which` of the `ntab` digit-addressed tables of this lane
__device__ __forceinline__ LaneTable lane_table_at(uint4 *ws, int which, int ntab) {
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    return LaneTable{ws + ((size_t)lane * ntab + which) * TABLE_U4};
}

// ================================================================== digit-addressed tables (public scalars)

// (One scalar times a variable base -- goldilocks_448_point_scalarmul, goldilocks_448_direct_scalarmul -- had a
// digit-addressed form here until round 6: a 16-entry table per resident lane in HBM, 544 MiB of workspace and 46 x the
// algorithmic traffic, for 32.4 M/s against the table-free ladder's 32.8 (direct_scalarmul: 26.6 against 28.5).  Both
// table-access modes now run the ladder (kernels_varbase_ct.hip), and so does goldilocks_448_point_dual_scalarmul, whose
// one table walked twice was no faster than two ladders either (62.7 against 61.7 ms); the per-lane tables stay where
// they win: two scalars on ONE doubling chain (point_double_scalarmul: 42 against 62 ms), and verification.)

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]   (ref: goldilocks_448_point_double_scalarmul, src/goldilocks.c:467-541).
// b1 == nullptr: b1 is the base point -- goldilocks_448_base_double_scalarmul_non_secret, public scalars by
// contract (src/goldilocks.c:1260-1330): s2*b2 by the one-table ladder, then s1*B as 28 additions from the base
// point's 16-bit window table onto the same accumulator (no doublings for the base point's half).
// out may alias b2.
__device__ __forceinline__ void double_scalarmul_body(uint64_t *out, const uint64_t *b1,
                                                      const uint64_t *__restrict__ s1, const uint64_t *b2,
                                                      const uint64_t *__restrict__ s2, uint32_t n,
                                                      uint4 *__restrict__ workspace,
                                                      const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_bits[30 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable t2 = lane_table_at(workspace, 0, 2), t1 = lane_table_at(workspace, 1, 2);
    for (uint32_t i = lane; i < n; i += stride) {
        LdsBits bits2 = lds_put_bits(s_bits + 15 * BLOCK + threadIdx.x, sc_recode_signed(sc_load_abi(s2 + 7 * (size_t)i)));
        build_window_table(t2, pt_load_abi(b2 + 32 * (size_t)i));
        pt r;
        if (b1) {  // uniform
            LdsBits bits1 = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(s1 + 7 * (size_t)i)));
            build_window_table(t1, pt_load_abi(b1 + 32 * (size_t)i));
            r = ladder_double(bits1, t1, bits2, t2);
        } else {
            r = ladder_varbase(bits2, t2);
            const GlobalBwt base_tab{bwt};
            LdsBits bits1 = lds_put_bits(s_bits + threadIdx.x, sc_recode_bwt(sc_load_abi(s1 + 7 * (size_t)i), base_tab));
            ladder_bwt_onto(r, bits1, base_tab);
        }
        pt_store_abi(out + 32 * (size_t)i, r);
    }
}

// ================================================================== no table: the Montgomery ladder (secret scalars)

// config 2, index-independent (the library default): scaled[i] = scalar[i] * base[i] by the table-free
// Montgomery ladder of montgomery.hpp   (ref: goldilocks_448_point_scalarmul, constant time there:
// src/goldilocks.c:437-442 through constant_time.h:134-183).
// The ladder wants u(P) = (Y+Z)/(Y-Z) in affine form: the inversions of the operations a lane owns are
// shared (InvChain, fixed_bodies.hpp: Montgomery's trick along the lane) -- a first pass over the lane's
// base points parks Y-Z and the running product (ML_SLOT_U4 uint4 per operation), one inversion, then
// the operations run in reverse order, re-reading their base point.  out may alias base: an operation's
// base point is read (again) right before its result is written, and the first pass is over by then.
template <class BODY>
__device__ __forceinline__ void for_each_wave_round_reverse(uint32_t n, BODY body) {
    const uint32_t l = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * BLOCK + threadIdx.x - l, stride = gridDim.x * BLOCK;
    if (first >= n) return;
    for (uint32_t i0 = first + (n - 1 - first) / stride * stride;; i0 -= stride) {
        body(i0);
        if (i0 < stride) break;
    }
}
__device__ __forceinline__ void point_scalarmul_ladder_body(uint64_t *out, const uint64_t *base,
                                                            const uint64_t *__restrict__ scalar, uint32_t n,
                                                            uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint4 s_stage[(BLOCK / 64) * WAVE_STAGE_U4];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t l = threadIdx.x & 63u;
    uint4 *stage = s_stage + (threadIdx.x >> 6) * WAVE_STAGE_U4;
    InvChain ch;
    ch.begin();
    for (uint32_t i0 = lane - l; i0 < n; i0 += stride) {   // wave-uniform: 64 consecutive operations per round
        const uint32_t m = n - i0 < 64u ? n - i0 : 64u;
        const pt b = wave_load_points(stage, base, i0, m, l);
        ch.push(workspace + (size_t)ML_SLOT_U4 * (i0 + l), ml_denominator(b), l < m);
    }
    ch.invert_wave(reinterpret_cast<uint32_t *>(stage), false);   // one inversion per wave, in the wave's own staging region
    for_each_wave_round_reverse(n, [&](uint32_t i0) {
        const uint32_t m = n - i0 < 64u ? n - i0 : 64u;
        const pt b = wave_load_points(stage, base, i0, m, l);
        const sc k = wave_load_scalars(stage, scalar, i0, m, l);
        pt r = b;
        if (l < m) {
            const fe di = ch.pop(workspace + (size_t)ML_SLOT_U4 * (i0 + l));
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_reduce(k));
            r = ml_scalarmul(b, di, bits);
        }
        wave_store_points(stage, out, i0, m, l, r);
    });
    // the scalar was secret: neither its words nor its staged copy stay in LDS
    lds_wipe_lane(s_bits + threadIdx.x, 15);
    wave_sync();
    for (int k = 0; k < WAVE_STAGE_U4 / 64; k++) stage[k * 64 + l] = make_uint4(0, 0, 0, 0);
}

// "next" row f2, index-independent: decode, ladder, encode.  The decoder hands out u(P) with the point (one
// exponentiation for both, point.hpp pt_decode_words_u), so this kernel needs no shared inversion and no workspace.
__device__ __forceinline__ void direct_scalarmul_ladder_body(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                                                             const uint8_t *__restrict__ base,
                                                             const uint64_t *__restrict__ scalar, uint32_t n,
                                                             int allow_identity, int short_circuit,
                                                             const uint64_t *__restrict__ point_base_abi) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = lane; i < n; i += stride) {
        uint32_t w[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        pt b;
        fe u;
        const bool ok = pt_decode_words_u(b, u, w, allow_identity != 0);
        status[i] = ok ? -1 : 0;
        if (!ok && short_circuit) continue;         // the encoding is public: so is this branch
        if (!ok) {                                  // src/goldilocks.c:898: multiply the base point instead
            b = pt_load_abi(point_base_abi);
#if defined(GD_REPRO_R05_DIVERGENT_INVERSION)
            // Round 5's miscompiled shape, kept for tools/probes/miscompile_r05_repro.py ONLY: the field inversion inside
            // this divergent block came out wrong in the no-pairs build of this unit (docs/history/r05.md H).
            u = fe_mul(fe_add(b.y, b.z), fe_invert(ml_denominator(b)));
#else
            u = ml_u_base();
#endif
        }
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_reduce(sc_load_abi(scalar + 7 * (size_t)i)));
        const pt r = ml_scalarmul_u(b, u, bits);
        pt_encode_words(w, r);
        uint32_t *dst = reinterpret_cast<uint32_t *>(scaled + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = w[k];
    }
    lds_wipe_lane(s_bits + threadIdx.x, 15);
}

// "next" row f4, index-independent: two ladders from one u(P); the inversion behind it is shared along the lane
// as in the headline kernel.  ML_DUAL_SLOT_U4 uint4 of workspace per operation: the chain's slot, which -- popped --
// holds s2 * P until s1 * P is there as well and both are written: out1 and out2 may alias base (the reference
// computes into temporaries, src/goldilocks.c:543-642).
__device__ __forceinline__ void point_dual_scalarmul_ladder_body(uint64_t *out1, uint64_t *out2,
                                                                 const uint64_t *base, const uint64_t *__restrict__ s1,
                                                                 const uint64_t *__restrict__ s2, uint32_t n,
                                                                 uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    InvChain ch;
    ch.begin();
    for_each_op<true>(n, [&](uint32_t i, bool live) GD_LAMBDA_INLINE {
        ch.push(workspace + (size_t)ML_DUAL_SLOT_U4 * i, ml_denominator(pt_load_abi(base + 32 * (size_t)i)), live);
    });
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    ch.invert_wave(s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS, false);
    // ONE copy of the ladder, walked twice (a loop the compiler may not unroll): two inlined copies keep the first
    // result and both recoveries' temporaries alive across each other -- 1 321 spilled registers, profiles/r03 --
    // where the single ladder of k_point_scalarmul_ct spills 77.
    for_each_op_reverse(n, [&](uint32_t i) GD_LAMBDA_INLINE {
        uint4 *slot = workspace + (size_t)ML_DUAL_SLOT_U4 * i;
        const fe di = ch.pop(slot);
#pragma unroll 1
        for (int which = 1; which >= 0; which--) {
            const pt b = pt_load_abi(base + 32 * (size_t)i);
            const uint64_t *k = (which ? s2 : s1) + 7 * (size_t)i;
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_reduce(sc_load_abi(k)));
            const pt r = ml_scalarmul(b, di, bits);
            if (which) {
                pt_store_u4(slot, r);
            } else {
                pt_store_abi(out1 + 32 * (size_t)i, r);
                pt_store_abi(out2 + 32 * (size_t)i, pt_load_u4(slot));
            }
        }
    });
    lds_wipe_lane(s_bits + threadIdx.x, 15);
}

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i], index-independent: ONE two-dimensional differential ladder (montgomery2d.hpp: a
// doubling and two differential additions per bit instead of two ladders' two and two -- 0.82 x the instructions; until
// round 6 this was two ladders and an addition).  First pass: the exceptional inputs are substituted (ml2_effective) and
// the denominators of u(p1), u(p2), u(p1 + p2), u(p1 - p2) go into the lane's chain; one inversion per wave; second pass:
// the four affine differences into LDS (pair-interleaved: both candidates of a selection are one 64-bit read), the
// scalars and their control stream into the operation's workspace slot (read a word per 32 steps; wiped afterwards),
// the chain, the recovery.  ML_DOUBLE_SLOT_U4 uint4 of workspace per operation: four chain slots | a, b, c.
// out is written when both base points have been read for the last time: it may alias either (the reference computes
// into temporaries, src/goldilocks.c:467-541).
struct LdsDiffs {
    uint32_t *mine;     // s_diff + 2 * threadIdx.x: word ((k * 16 + limb) * BLOCK + tid) * 2 + which
    __device__ __forceinline__ void put(int k, int which, const fe &v) const {
#pragma unroll
        for (int i = 0; i < 16; i++) mine[((k * 16 + i) * BLOCK) * 2 + which] = v.v[i];
    }
    __device__ __forceinline__ sfp pair(int k, bool second) const {      // both read, one kept: no address depends on `second`
        // (read HERE, every time: left to itself the compiler reads the 64 words once, ahead of the chain's loop, finds no
        // registers for them and re-reads them from SCRATCH in every step -- 33 reads and 11 waits per step, round 6)
        uint32_t at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)mine;
        asm volatile("" : "+v"(at));
        sfp r;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint64_t both = *(const __attribute__((address_space(3))) uint64_t *)(uintptr_t)(at + 8u * (uint32_t)((k * 16 + i) * BLOCK));
            r.v[i] = (int32_t)(second ? (uint32_t)(both >> 32) : (uint32_t)both);
            asm("" : "+v"(r.v[i]));     // opaque, like sfe_from_fe: no zero-extended multiplicands
        }
        return r;
    }
};
struct GlobalWords {    // a scalar's words in the operation's workspace slot
    const uint32_t *p;
    __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
};
__device__ __forceinline__ fe ml_u_numerator(const pt &b) { return fe_weak(fe_add(b.y, b.z)); }      // u = (Y + Z) / (Y - Z)
// The operation's workspace slot (ML_DOUBLE_SLOT_U4 uint4): everything that depends on the POINTS is decided in the first
// pass and parked, so that the second pass is the chain and little else (with the points' logic inlined there as well
// the kernel spilled 1 400 registers):
//   [0, 32)   four chain slots: the denominators Y - Z of p1, p2, p1 + p2, p1 - p2 (ml2_effective's points)
//   [32, 48)  the numerators Y + Z of the same four
//   [48, 64)  p1 (for the recovery)
//   [64, 76)  the scalars a, b and -- second pass -- their control stream c: 3 x 14 words; wiped when the operation is done
constexpr int ML2_NUM_AT = 4 * ML_SLOT_U4, ML2_P1_AT = ML2_NUM_AT + 16, ML2_WORDS_AT = ML2_P1_AT + 16;
static_assert(ML2_WORDS_AT + 12 == ML_DOUBLE_SLOT_U4, "the slot's layout");
__device__ __forceinline__ void double_scalarmul_ladder_body(uint64_t *out, const uint64_t *b1,
                                                             const uint64_t *__restrict__ s1, const uint64_t *b2,
                                                             const uint64_t *__restrict__ s2, uint32_t n,
                                                             uint4 *__restrict__ workspace,
                                                             const uint64_t *__restrict__ point_base_abi) {
    // the differences' 64 KiB double as the wave inversions' staging between the passes
    __shared__ uint32_t s_diff[64 * BLOCK];
    static_assert(64 * BLOCK >= (BLOCK / 64) * INV_WAVE_LDS_WORDS, "the inversions fit the differences' region");
    InvChain ch;
    ch.begin();
    for_each_op<true>(n, [&](uint32_t i, bool live) GD_LAMBDA_INLINE {
        const Ml2Inputs in = ml2_effective(pt_load_abi(b1 + 32 * (size_t)i), pt_load_abi(b2 + 32 * (size_t)i),
                                           sc_reduce(sc_load_abi(s1 + 7 * (size_t)i)), sc_reduce(sc_load_abi(s2 + 7 * (size_t)i)),
                                           pt_load_abi(point_base_abi));
        uint4 *slot = workspace + (size_t)ML_DOUBLE_SLOT_U4 * i;
        const pt sum = pt_add(in.p1, in.p2, false), dif = pt_add(in.p1, in.p2, true);
        if (live) {
            fe_store(slot + ML2_NUM_AT, ml_u_numerator(in.p1));
            fe_store(slot + ML2_NUM_AT + 4, ml_u_numerator(in.p2));
            fe_store(slot + ML2_NUM_AT + 8, ml_u_numerator(sum));
            fe_store(slot + ML2_NUM_AT + 12, ml_u_numerator(dif));
            pt_store_u4(slot + ML2_P1_AT, in.p1);
            uint32_t *words = reinterpret_cast<uint32_t *>(slot + ML2_WORDS_AT);
#pragma unroll
            for (int k = 0; k < 14; k++) {
                words[k] = in.a.w[k];
                words[14 + k] = in.b.w[k];
            }
        }
        ch.push(slot, ml_denominator(in.p1), live);
        ch.push(slot + ML_SLOT_U4, ml_denominator(in.p2), live);
        ch.push(slot + 2 * ML_SLOT_U4, ml_denominator(sum), live);
        ch.push(slot + 3 * ML_SLOT_U4, ml_denominator(dif), live);
    });
    __syncthreads();
    ch.invert_wave(s_diff + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS, false);
    __syncthreads();
    const LdsDiffs diffs{s_diff + 2 * threadIdx.x};
    for_each_op_reverse(n, [&](uint32_t i) GD_LAMBDA_INLINE {
        uint4 *slot = workspace + (size_t)ML_DOUBLE_SLOT_U4 * i;
        uint32_t *words = reinterpret_cast<uint32_t *>(slot + ML2_WORDS_AT);       // a | b | c, 14 words each
        // the chain pops in the reverse order of its pushes
        diffs.put(1, 1, fe_mul(fe_load(slot + ML2_NUM_AT + 12), ch.pop(slot + 3 * ML_SLOT_U4)));
        diffs.put(1, 0, fe_mul(fe_load(slot + ML2_NUM_AT + 8), ch.pop(slot + 2 * ML_SLOT_U4)));
        diffs.put(0, 1, fe_mul(fe_load(slot + ML2_NUM_AT + 4), ch.pop(slot + ML_SLOT_U4)));
        const fe u1 = fe_mul(fe_load(slot + ML2_NUM_AT), ch.pop(slot));
        diffs.put(0, 0, u1);
        {
            sc a, b;
            uint32_t c[14];
#pragma unroll
            for (int k = 0; k < 14; k++) {
                a.w[k] = words[k];
                b.w[k] = words[14 + k];
            }
            ml2_control(c, a, b);
#pragma unroll
            for (int k = 0; k < 14; k++) words[28 + k] = c[k];
        }
        const GlobalWords wa{words}, wb{words + 14}, wc{words + 28};
        const pt r = ml2_double_scalarmul_core(u1, wa, wb, wc, diffs, [&]() GD_LAMBDA_INLINE { return pt_load_u4(slot + ML2_P1_AT); });
#pragma unroll
        for (int k = 0; k < 42; k++) words[k] = 0;      // the scalars do not stay behind in the workspace
        pt_store_abi(out + 32 * (size_t)i, r);
    });
}

}  // namespace gd
