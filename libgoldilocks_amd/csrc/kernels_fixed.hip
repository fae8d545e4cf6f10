// kernels_fixed.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "fixed_bodies.hpp"

namespace gd {

// config 3: scaled[i] = scalar[i] * G, G given by a comb table   (ref: goldilocks_448_precomputed_scalarmul)
GD_KERNEL k_precomputed_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ comb,
                                  const uint64_t *__restrict__ scalar, uint32_t n) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    __shared__ uint4 s_stage[(BLOCK / 64) * WAVE_STAGE_U4];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t l = threadIdx.x & 63u;
    uint4 *stage = s_stage + (threadIdx.x >> 6) * WAVE_STAGE_U4;
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, l};
    // every lane of a wave must take part in the shuffles: the loop is wave-uniform and lanes past
    // the end multiply by a zero scalar (their result is not stored)
    for (uint32_t i0 = lane - l; i0 < n; i0 += stride) {
        const uint32_t m = n - i0 < 64u ? n - i0 : 64u;
        const sc k = wave_load_scalars(stage, scalar, i0, m, l);
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(k));
        pt res = ladder_comb(bits, tab);
        wave_store_points(stage, out, i0, m, l, res);
    }
    // the scalar may have been secret: neither its recoding nor its staged copy stays in LDS
    lds_wipe_lane(s_bits + threadIdx.x, 15);
    wave_sync();
    for (int k = 0; k < WAVE_STAGE_U4 / 64; k++) stage[k * 64 + l] = make_uint4(0, 0, 0, 0);
}

// scaled[i] = scalar[i] * B for the built-in base point with index-independent table access: the library's
// own 4 x 7 x 16 comb, staged in LDS and gathered with wavefront shuffles (kernels.hpp LdsShuffleCombBig)
// halve != 0: the comb is one of 2G (k_recomb_big, a caller's table re-combed): the scalar is halved first.
GD_KERNEL k_base_scalarmul_ct(uint64_t *__restrict__ out, const uint4 *__restrict__ comb_big_tab,
                              const uint64_t *__restrict__ scalar, uint32_t n, uint32_t halve) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_BIG_LDS_WORDS];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    stage_comb_lds<comb_big::ENTRIES>(s_comb, comb_big_tab);
    LdsShuffleCombBig tab{s_comb, threadIdx.x & 63u};
    // every lane of a wave must take part in the shuffles: the loop is wave-uniform and lanes past the end
    // redo the last operation without storing it
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i = lane + r * stride;
        const uint32_t j = i < n ? i : n - 1;
        sc k = sc_load_abi(scalar + 7 * (size_t)j);
        if (halve) k = sc_halve(k);   // uniform
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, comb_big::recode(k));
        const pt res = ladder_comb(bits, tab);
        if (i < n) pt_store_abi(out + 32 * (size_t)i, res);
    }
    lds_wipe_lane(s_bits + threadIdx.x, 15);   // the scalar may have been secret
}

// scaled[i] = scalar[i] * B for the built-in base point, through the window table
GD_KERNEL k_base_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ bwt,
                           const uint64_t *__restrict__ scalar, uint32_t n) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint4 s_stage[(BLOCK / 64) * WAVE_STAGE_U4];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t l = threadIdx.x & 63u;
    uint4 *stage = s_stage + (threadIdx.x >> 6) * WAVE_STAGE_U4;
    GlobalBwt tab{bwt};
    for (uint32_t i0 = lane - l; i0 < n; i0 += stride) {
        const uint32_t m = n - i0 < 64u ? n - i0 : 64u;
        const sc k = wave_load_scalars(stage, scalar, i0, m, l);
        pt res = pt_identity();
        if (l < m) {
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_bwt(k, tab));
            res = ladder_bwt(bits, tab);
        }
        wave_store_points(stage, out, i0, m, l, res);
    }
}

// ---- the base point's window table: T_i[k] = (2k+1) * 2^(bits i) * B as affine niels ((Y-X)/2Z, (Y+X)/2Z, 78164 T/2Z).
// The scalar m * 2^(bits i) mod q, m < 2^bits: shifted as far as 448 bits allow, reduced, the rest doubled.
__device__ __forceinline__ sc bwt_entry_scalar(uint32_t m, uint32_t bits, uint32_t i) {
    sc v = sc_zero();
    uint32_t bit = bits * i, extra = 0;
    if (bit + bits > 448) {
        extra = bit + bits - 448;
        bit -= extra;
    }
    const uint32_t wd = bit >> 5, sh = bit & 31;
#pragma unroll
    for (int w = 0; w < 14; w++) {
        uint32_t x = 0;
        if ((uint32_t)w == wd) x = m << sh;
        if ((uint32_t)w == wd + 1 && sh + bits > 32) x = m >> (32 - sh);
        v.w[w] = x;
    }
    v = sc_reduce(v);
    for (uint32_t k = 0; k < extra; k++) v = sc_add(v, v);
    return v;
}
GD_KERNEL k_bwt_header(uint4 *__restrict__ table, uint32_t bits) {
    if (blockIdx.x || threadIdx.x >= (uint32_t)BWT_HEADER_U4) return;
    const sc a = bwt_adjust_for(bits);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) v = make_uint4(bits, bwt_windows(bits), 0, 0);
    if (threadIdx.x == 1) v = make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]);
    if (threadIdx.x == 2) v = make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]);
    if (threadIdx.x == 3) v = make_uint4(a.w[8], a.w[9], a.w[10], a.w[11]);
    if (threadIdx.x == 4) v = make_uint4(a.w[12], a.w[13], 0, 0);
    table[threadIdx.x] = v;
}
// steps[i] = 2 * 2^(bits i) * B as a projective niels: what takes an entry of window i to the next.  One block (the
// comb gather needs full waves: lanes beyond the windows repeat the last one).
GD_KERNEL k_bwt_steps(uint4 *__restrict__ steps, const uint4 *__restrict__ comb, uint32_t bits) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    const uint32_t windows = bwt_windows(bits), i = threadIdx.x < windows ? threadIdx.x : windows - 1;
    LdsBits b = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(bwt_entry_scalar(1u, bits, i)));
    pt p = ladder_comb(b, tab);
    pt_double(p, true);
    if (!blockIdx.x && threadIdx.x < windows) pniels_store(steps + 16 * i, pt_to_pniels(p));
}
// A lane owns BWT_BUILD_SEG consecutive entries of one window: the first from the comb (a whole scalar multiplication),
// each further one by adding the window's step, all normalised with ONE shared inversion (Montgomery's trick along the
// lane, as k_verify_key_combs): 34 multiplications per entry instead of the 1 200 of an entry per lane with its own
// multiplication and inversion -- 2^23 x 19 entries in tens of milliseconds.  The unnormalised entries wait in their
// own slots; chain: 8 uint4 per entry of the slab [first, first + count) (both multiples of BWT_BUILD_SEG).
GD_KERNEL k_build_bwt(uint4 *__restrict__ table, const uint4 *__restrict__ comb, const uint4 *__restrict__ steps,
                      uint4 *__restrict__ chain, uint32_t bits, uint64_t first, uint32_t count) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = (uint64_t)t * BWT_BUILD_SEG < count;
    const uint32_t off = live ? t * BWT_BUILD_SEG : 0u;          // (idle lanes of the last block repeat the first segment)
    const uint64_t e0 = first + off;
    const uint32_t i = (uint32_t)(e0 >> (bits - 1)), k0 = (uint32_t)e0 & ((1u << (bits - 1)) - 1);
    LdsBits b = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(bwt_entry_scalar(2 * k0 + 1, bits, i)));
    pt p = ladder_comb(b, tab);
    const pniels step = pniels_load(steps + 16 * i);
    uint4 *const entries = table + BWT_HEADER_U4 + 12 * e0;
    uint4 *const slots = chain + 8 * (size_t)off;
    InvChain ch;
    ch.begin();
#pragma unroll 1
    for (uint32_t s = 0;; s++) {
        uint4 *q = entries + 12 * s;
        if (live) {
            fe_store(q, fe_weak(fe_sub<2>(p.y, p.x)));
            fe_store(q + 4, fe_weak(fe_add(p.x, p.y)));
            fe_store(q + 8, fe_mulw(p.t, TWO_EFF_D));
        }
        ch.push(slots + 8 * s, fe_add(p.z, p.z), live);
        if (s + 1 == BWT_BUILD_SEG) break;
        pt_add_pniels(p, step, false, true);
    }
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    ch.invert_wave(s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS, false);   // one inversion per wave (inv_wave.hpp)
    if (!live) return;
#pragma unroll 1
    for (uint32_t s = BWT_BUILD_SEG; s-- > 0;) {
        const fe zi = ch.pop(slots + 8 * s);
        uint4 *q = entries + 12 * s;
        fe_store(q, fe_mul(fe_load(q), zi));
        fe_store(q + 4, fe_mul(fe_load(q + 4), zi));
        fe_store(q + 8, fe_mul(fe_load(q + 8), zi));
    }
}

// entry e = 64 j + idx of the 4 x 7 x 16 comb: (2^(16(6+7j)) + sum_{k<6} (+-) 2^(16(k+7j))) * B, + iff bit k of idx,
// as affine niels in our form; computed like the window table above.  Launch with exactly comb_big::ENTRIES lanes.
GD_KERNEL k_build_comb_big(uint4 *__restrict__ dst, const uint4 *__restrict__ comb) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    const uint32_t e = blockIdx.x * BLOCK + threadIdx.x;   // < comb_big::ENTRIES by construction
    const uint32_t j = e / comb_big::PER_COMB, idx = e % comb_big::PER_COMB;
    const auto power = [](uint32_t bit) {
        sc v = sc_zero();
#pragma unroll
        for (int w = 0; w < 14; w++) v.w[w] = (uint32_t)w == (bit >> 5) ? 1u << (bit & 31) : 0u;
        return v;
    };
    sc v = power(comb_big::SPACING * (comb_big::TEETH - 1 + comb_big::TEETH * j));
#pragma unroll 1
    for (uint32_t k = 0; k + 1 < (uint32_t)comb_big::TEETH; k++) {
        const sc term = power(comb_big::SPACING * (k + comb_big::TEETH * j));
        v = (idx >> k) & 1u ? sc_add(v, term) : sc_sub(v, term);
    }
    LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(v));
    pt p = ladder_comb(bits, tab);
    fe zi = fe_invert(fe_weak(fe_add(p.z, p.z)));
    uint4 *q = dst + 12 * (size_t)e;
    fe_store(q, fe_mul(fe_weak(fe_sub<2>(p.y, p.x)), zi));
    fe_store(q + 4, fe_mul(fe_weak(fe_add(p.x, p.y)), zi));
    fe_store(q + 8, fe_mul(fe_mulw(p.t, TWO_EFF_D), zi));
}

// A caller's 5 x 5 x 18 comb (our form, k_import_comb) -> the 4 x 7 x 16 comb of TWICE its base point, without a
// single scalar multiplication: the reference comb's entries are signed sums of the teeth T_a = 2^(18a) G, so
//     entry(j, 1 << k) - entry(j, 0) = 2 T_(k+5j)  (k < 4),      entry(j, 15) + entry(j, 0) = 2 T_(4+5j);
// tooth m of the new comb, 2^(16m) * 2G, is 2 T_a doubled 16m - 18a <= 17 times (28 lanes), every entry is
// the signed sum of 7 teeth (256 lanes, 6 additions) and one inversion per lane makes it an affine niels.
// About 0.4 ms of latency on one block: worth it from 2^18 operations on (goldilocks_amd.hip).  The base
// is 2G, so the multiplying kernel halves its scalars.  Launch with exactly one block of comb_big::ENTRIES lanes.
GD_KERNEL k_recomb_big(uint4 *__restrict__ dst, const uint4 *__restrict__ comb) {
    static_assert(comb_big::ENTRIES == BLOCK, "one lane per entry, one block");
    __shared__ uint4 s_teeth[comb_big::TEETH * comb_big::COMBS * 16];
    const uint32_t e = threadIdx.x;
    const auto ref_entry = [&](uint32_t j, uint32_t idx) {
        const uint4 *q = comb + 12 * (16 * j + idx);
        niels n;
        n.a = fe_load(q);
        n.b = fe_load(q + 4);
        n.cn = fe_load(q + 8);
        return n;
    };
    if (e < (uint32_t)(comb_big::TEETH * comb_big::COMBS)) {
        const uint32_t bit = comb_big::SPACING * e, a = bit / 18, extra = bit - 18 * a, j = a / 5, k = a % 5;
        pt t = niels_to_pt(ref_entry(j, k < 4 ? 1u << k : 15u), false);
        pt_add_niels(t, ref_entry(j, 0), k < 4, true);              // 2 T_a
#pragma unroll 1
        for (uint32_t d = 0; d < extra; d++) pt_double(t, true);
        pniels_store(s_teeth + 16 * e, pt_to_pniels(t));
    }
    __syncthreads();
    niels_store(dst + 12 * (size_t)e, comb_big_entry(LdsTeeth{s_teeth}, e));
}

// enc[i] = RFC 8032 encoding of 4 * pts[i]   (ref: goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa,
// src/goldilocks.c:905-946) for large batches: the one field inversion per point -- five sixths of the work --
// is shared between the points a lane handles (InvChain, as in key derivation).
// workspace: DERIVE_SLOT_U4 uint4 per point (xn | yn | denominator | prefix)
GD_KERNEL k_point_encode_eddsa_shared(uint8_t *__restrict__ enc, const uint64_t *__restrict__ pts, uint32_t n,
                                      uint4 *__restrict__ ws) {
    InvChain ch;
    ch.begin();
    for_each_op<false>(n, [&](uint32_t i, bool) {
        fe xn, yn, zn;
        pt_eddsa_isogeny(xn, yn, zn, pt_load_abi(pts + 32 * (size_t)i));
        uint4 *slot = ws + (size_t)DERIVE_SLOT_U4 * i;
        fe_store(slot, xn);
        fe_store(slot + 4, yn);
        ch.push(slot + 8, zn, true);
    });
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    ch.invert_wave(s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS, false);
    for_each_op_reverse(n, [&](uint32_t i) {
        const uint4 *slot = ws + (size_t)DERIVE_SLOT_U4 * i;
        const fe zi = ch.pop(slot + 8);
        uint32_t w[15];
        eddsa_finish_words(w, fe_load(slot), fe_load(slot + 4), zi);
        store_words_as_bytes(enc + 57 * (size_t)i, w, 57);
    });
}

GD_KERNEL k_ed448_derive_public_key(uint8_t *__restrict__ pk, const uint8_t *__restrict__ sk, uint32_t n,
                                    const uint4 *__restrict__ bwt, uint4 *__restrict__ workspace) {
    derive_body<false>(pk, sk, n, bwt, workspace);
}

GD_KERNEL k_ed448_sign(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                       const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                       uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                       uint32_t n, const uint4 *__restrict__ bwt, uint4 *__restrict__ workspace) {
    sign_body<false>(sig, sk, pk, msgs, msg_offsets, msg_len, prehashed, ctx, ctx_len, n, bwt, workspace);
}

GD_KERNEL k_x448(uint8_t *__restrict__ shared, int32_t *__restrict__ status, const uint8_t *__restrict__ base,
                 const uint8_t *__restrict__ scalar, uint32_t n, const uint4 *__restrict__ bwt,
                 uint4 *__restrict__ workspace) {
    x448_body<false>(shared, status, base, scalar, n, bwt, workspace);
}

}  // namespace gd
