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
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_bwt(k));
            res = ladder_bwt(bits, tab);
        }
        wave_store_points(stage, out, i0, m, l, res);
    }
}

// T_i[k] = (2k+1) * 2^(BWT_BITS*i) * B as affine niels ((Y-X)/2Z, (Y+X)/2Z, 78164 T/2Z), one entry per lane.
// Launch with exactly BWT_ENTRIES lanes (a multiple of the block size): the comb gather needs full waves.
GD_KERNEL k_build_bwt(uint4 *__restrict__ dst, const uint4 *__restrict__ comb) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    const uint32_t e = blockIdx.x * BLOCK + threadIdx.x;   // < BWT_ENTRIES by construction
    const uint32_t i = e / BWT_PER_WINDOW, kk = e % BWT_PER_WINDOW;
    // the scalar (2k+1) * 2^(BWT_BITS*i) mod q: shift as far as 448 bits allow, reduce, double the rest
    sc v = sc_zero();
    const uint32_t m = 2 * kk + 1;               // < 2^BWT_BITS
    uint32_t bit = BWT_BITS * i, extra = 0;
    if (bit + BWT_BITS > 448) {
        extra = bit + BWT_BITS - 448;
        bit -= extra;
    }
    const uint32_t wd = bit >> 5, sh = bit & 31;
#pragma unroll
    for (int w = 0; w < 14; w++) {
        uint32_t x = 0;
        if ((uint32_t)w == wd) x = m << sh;
        if ((uint32_t)w == wd + 1 && sh + BWT_BITS > 32) x = m >> (32 - sh);
        v.w[w] = x;
    }
    v = sc_reduce(v);
    for (uint32_t k = 0; k < extra; k++) v = sc_add(v, v);
    LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(v));
    pt p = ladder_comb(bits, tab);
    fe zi = fe_invert(fe_weak(fe_add(p.z, p.z)));
    uint4 *q = dst + 12 * (size_t)e;
    fe_store(q, fe_mul(fe_weak(fe_sub<2>(p.y, p.x)), zi));
    fe_store(q + 4, fe_mul(fe_weak(fe_add(p.x, p.y)), zi));
    fe_store(q + 8, fe_mul(fe_mulw(p.t, TWO_EFF_D), zi));
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
    ch.invert();
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
