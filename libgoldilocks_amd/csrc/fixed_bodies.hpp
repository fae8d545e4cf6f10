// fixed_bodies.hpp -- bodies of the kernels that multiply the base point by a secret scalar, as
// templates on the table-access policy; instantiated by kernels_fixed.hip (CT = false) and
// kernels_fixed_ct.hip (CT = true) so the two sets compile in parallel.
#pragma once
#include "kernels.hpp"

namespace gd {

// ---- fixed-base policy of the kernels below.  CT = false: the base point's 8-bit window table in
// global memory (fast, table addresses depend on the digit).  CT = true: the 5x5x18 comb staged in
// LDS with the wavefront-shuffle gather (addresses and timing independent of the secret digit --
// the counterpart of the reference's constant_time_lookup, constant_time.h:61-362).  The shuffle
// needs every lane of a wave, so the CT loops run wave-uniformly and idle lanes redo the last
// operation (they store the same bytes to the same place).
template <bool CT>
__device__ __forceinline__ uint32_t *fixed_base_stage(const uint4 *table) {
    if constexpr (CT) {
        __shared__ uint32_t s_comb[COMB_LDS_WORDS];
        stage_comb_lds(s_comb, table);
        return s_comb;
    } else {
        return nullptr;
    }
}
// for (op = every operation this lane owns) body(index)
template <bool CT, class BODY>
__device__ __forceinline__ void for_each_op(uint32_t n, BODY body) {
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    if constexpr (CT) {
        const uint32_t rounds = (n + stride - 1) / stride;
        for (uint32_t r = 0; r < rounds; r++) {
            const uint32_t i = lane + r * stride;
            body(i < n ? i : n - 1);
        }
    } else {
        for (uint32_t i = lane; i < n; i += stride) body(i);
    }
}

// "next" row f1: pk[i] = derive_public_key(sk[i])   (ref: goldilocks_ed448_derive_public_key)
template <bool CT>
__device__ __forceinline__ void derive_body(uint8_t *pk, const uint8_t *sk, uint32_t n, const uint4 *table) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    uint32_t *s_comb = fixed_base_stage<CT>(table);
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    if constexpr (CT) {
        LdsShuffleComb comb{s_comb, threadIdx.x & 63u};
        FixedComb<LdsShuffleComb> fb{comb};
        for_each_op<CT>(n, [&](uint32_t i) { ed448_derive_core(pk + 57 * (size_t)i, sk + 57 * (size_t)i, fb, stage, mk); });
    } else {
        GlobalBwt bwt_tab{table};
        FixedBwt<GlobalBwt> fb{bwt_tab};
        for_each_op<CT>(n, [&](uint32_t i) { ed448_derive_core(pk + 57 * (size_t)i, sk + 57 * (size_t)i, fb, stage, mk); });
    }
}

// "next" row f1: sig[i] = sign(sk[i], pk[i], msg[i])   (ref: goldilocks_ed448_sign)
template <bool CT>
__device__ __forceinline__ void sign_body(uint8_t *sig, const uint8_t *sk, const uint8_t *pk, const uint8_t *msgs,
                                          const uint64_t *msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                          const uint8_t *ctx, uint32_t ctx_len, uint32_t n, const uint4 *table,
                                          uint8_t *workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    uint32_t *s_comb = fixed_base_stage<CT>(table);
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    // the hashed-key seed of the signature in flight
    uint8_t *scratch = workspace + (size_t)(blockIdx.x * BLOCK + threadIdx.x) * 64;
    auto one = [&](uint32_t i, const auto &fb) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint32_t mlen = msg_offsets ? (uint32_t)(msg_offsets[i + 1] - msg_offsets[i]) : msg_len;
        ed448_sign_core(sig + 114 * (size_t)i, sk + 57 * (size_t)i, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx,
                        ctx_len, scratch, fb, stage, mk);
    };
    if constexpr (CT) {
        LdsShuffleComb comb{s_comb, threadIdx.x & 63u};
        FixedComb<LdsShuffleComb> fb{comb};
        for_each_op<CT>(n, [&](uint32_t i) { one(i, fb); });
    } else {
        GlobalBwt bwt_tab{table};
        FixedBwt<GlobalBwt> fb{bwt_tab};
        for_each_op<CT>(n, [&](uint32_t i) { one(i, fb); });
    }
}

// "next" row f3: X448.  base == nullptr: derive_public_key through the fixed-base table
// (ref: goldilocks_x448, goldilocks_x448_derive_public_key)
template <bool CT>
__device__ __forceinline__ void x448_body(uint8_t *shared, int32_t *status, const uint8_t *base,
                                          const uint8_t *scalar, uint32_t n, const uint4 *table) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    uint32_t *s_comb = fixed_base_stage<CT>(table);
    for_each_op<CT>(n, [&](uint32_t i) {
        uint32_t w[14], o[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(scalar + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        bool ok = true;
        if (base) {   // Montgomery ladder: no table, conditional swaps by select
            uint32_t b[14];
            const uint32_t *bs = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
            for (int k = 0; k < 14; k++) b[k] = bs[k];
            sc raw;
#pragma unroll
            for (int k = 0; k < 14; k++) raw.w[k] = w[k];
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, raw);
            ok = x448_core(o, b, bits);
        } else if constexpr (CT) {
            LdsShuffleComb comb{s_comb, threadIdx.x & 63u};
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(x448_public_scalar(w)));
            pt_encode_x448_words(o, ladder_comb(bits, comb));
        } else {
            GlobalBwt tab{table};
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed8(x448_public_scalar(w)));
            pt_encode_x448_words(o, ladder_bwt(bits, tab));
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(shared + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = o[k];
        if (status) status[i] = ok ? -1 : 0;
    });
}

}  // namespace gd
