// fixed_bodies.hpp -- bodies of the kernels that multiply the base point by a secret scalar, as
// templates on the table-access policy; instantiated by kernels_fixed.hip (CT = false) and
// kernels_fixed_ct.hip (CT = true) so the two sets compile in parallel.
#pragma once
#include "kernels.hpp"
#include "inv_wave.hpp"

namespace gd {

// ---- fixed-base policy of the kernels below.  CT = false: the base point's 16-bit window table in
// global memory (fast, table addresses depend on the digit).  CT = true: the 5x5x18 comb staged in
// LDS with the wavefront-shuffle gather (addresses and timing independent of the secret digit --
// the counterpart of the reference's constant_time_lookup, constant_time.h:61-362).  The shuffle
// needs every lane of a wave, so the CT loops run wave-uniformly: an idle lane of the last round
// recomputes the last operation with live = false and stores nothing.
template <bool CT>
__device__ __forceinline__ uint32_t *fixed_base_stage(const uint4 *table) {
    if constexpr (CT) {   // the library's own 4 x 7 x 16 comb of the base point (k_build_comb_big)
        __shared__ uint32_t s_comb[COMB_BIG_LDS_WORDS];
        stage_comb_lds<comb_big::ENTRIES>(s_comb, table);
        return s_comb;
    } else {
        return nullptr;
    }
}
// Key derivation and signing hash (a 136-byte SHAKE block staged per lane, 34 KiB per block) and multiply the
// base point (the 48-KiB comb) by turns, never at once: the two take turns in ONE LDS region, so that two
// blocks still fit a CU.  Every lane of the block calls mul() the same number of times in uniform control
// flow (for_each_op<true>), which makes the barriers legal: the first one says every lane is done hashing,
// the last one that nobody still gathers from the comb.  Restaging costs 48 cached loads per lane per call.
static_assert(COMB_BIG_LDS_WORDS >= 34 * BLOCK, "the SHAKE stage lives inside the comb's region");
struct RestagedCombBig {
    using plan = comb_big;
    uint32_t *region;
    const uint4 *table;
    template <class MK>
    __device__ __forceinline__ pt mul(const sc &s, MK &mk) const {
        auto bits = mk(plan::recode(s), 0);
        __syncthreads();
        stage_comb_lds<comb_big::ENTRIES>(region, table);   // ends with a barrier
        LdsShuffleCombBig comb{region, threadIdx.x & 63u};
        const pt r = ladder_comb(bits, comb);
        __syncthreads();
        return r;
    }
    template <class MK>
    __device__ __forceinline__ void add_to(pt &acc, const sc &s, MK &mk) const { acc = pt_add(acc, mul(s, mk), false); }
};

// body(index, live) for every operation this lane owns, in ascending / descending order
template <bool CT, class BODY>
__device__ __forceinline__ void for_each_op(uint32_t n, BODY body) {
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    if constexpr (CT) {
        const uint32_t rounds = (n + stride - 1) / stride;
        for (uint32_t r = 0; r < rounds; r++) {
            const uint32_t i = lane + r * stride;
            body(i < n ? i : n - 1, i < n);
        }
    } else {
        for (uint32_t i = lane; i < n; i += stride) body(i, true);
    }
}
template <class BODY>
__device__ __forceinline__ void for_each_op_reverse(uint32_t n, BODY body) {   // live operations only
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    if (lane >= n) return;
    for (uint32_t i = lane + (n - 1 - lane) / stride * stride;; i -= stride) {
        body(i);
        if (i < stride) break;
    }
}

// ---- Montgomery's trick along the operations one lane handles back to back.
// Key derivation, signing and X448 each end in one field inversion per operation (the affine
// coordinates of the EdDSA encoding, the ladder's x/z): 446 squarings, a third of a signature.  The
// inversions of the K operations a lane owns are independent, so the lane multiplies the
// denominators up in a first pass -- parking each denominator and the running product before it in
// the workspace -- inverts once, and walks back:
//     1/z_k = (z_0..z_k)^-1 * (z_0..z_{k-1}),        (z_0..z_{k-1})^-1 = (z_0..z_k)^-1 * z_k.
// A zero denominator is parked as 1 with prefix 0: that yields 1/z = 0 (gf_invert(0) = 0 in the
// reference, src/goldilocks.c:69-80) and leaves the chain intact.  The same identity the reference
// uses across table entries in gf_batch_invert (src/goldilocks.c:703-726).
struct InvChain {
    fe acc;
    __device__ __forceinline__ void begin() { acc = fe_one(); }
    // slot: 8 uint4 = denominator | product of the denominators before it
    __device__ __forceinline__ void push(uint4 *slot, const fe &z, bool live) {
        const bool zero = fe_is_zero(z);
        const fe ze = fe_select(fe_weak(z), fe_one(), zero);
        if (live) {
            fe_store(slot, ze);
            fe_store(slot + 4, fe_select(acc, fe_zero(), zero));
        }
        acc = fe_select(acc, fe_mul(acc, ze), live);
    }
    __device__ __forceinline__ void invert() { acc = fe_invert(acc); }
    // ... with ONE inversion for the 64 chains of the wave (inv_wave.hpp): every lane of the wave must call; lds = a
    // region of INV_WAVE_LDS_WORDS words that this wave alone uses while the call lasts.  wipe: the chains' products
    // belong to secret data (a nonce's or a secret scalar's multiple in projective form): nothing of them stays in LDS.
    __device__ __forceinline__ void invert_wave(uint32_t *lds, bool wipe) {
#if defined(GD_NO_WAVE_INVERT)   // (A/B and bisecting: every lane inverts for itself, as until round 5)
        (void)lds; (void)wipe;
        invert();
        return;
#endif
        acc = wave_shared_invert(acc, lds);
        if (wipe) {
            const uint32_t l = threadIdx.x & 63u;
#pragma unroll
            for (int i = 0; i < 16; i++) lds[l * 16 + i] = 0;
        }
    }
    __device__ __forceinline__ fe pop(const uint4 *slot) {   // in the reverse order of push
        const fe ze = fe_load(slot), pre = fe_load(slot + 4);
        const fe zi = fe_mul(acc, pre);
        acc = fe_mul(acc, ze);
        return zi;
    }
};
__device__ __forceinline__ void sc_store_u4(uint4 *p, const sc &s) {
    p[0] = make_uint4(s.w[0], s.w[1], s.w[2], s.w[3]);
    p[1] = make_uint4(s.w[4], s.w[5], s.w[6], s.w[7]);
    p[2] = make_uint4(s.w[8], s.w[9], s.w[10], s.w[11]);
    p[3] = make_uint4(s.w[12], s.w[13], 0, 0);
}
__device__ __forceinline__ sc sc_load_u4(const uint4 *p) {
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    sc s;
    s.w[0] = a.x; s.w[1] = a.y; s.w[2] = a.z; s.w[3] = a.w;
    s.w[4] = b.x; s.w[5] = b.y; s.w[6] = b.z; s.w[7] = b.w;
    s.w[8] = c.x; s.w[9] = c.y; s.w[10] = c.z; s.w[11] = c.w;
    s.w[12] = d.x; s.w[13] = d.y;
    return s;
}

// The chains' one inversion per wave (InvChain::invert_wave) in a staging region that the BLOCK shares (the SHAKE staging
// is word-interleaved across the block's lanes, the comb is the block's): every wave is done with the region before any
// wave's inversion writes to it, and every inversion is done before the second pass stages its hashes again.
__device__ __forceinline__ void block_invert_in_stage(InvChain &ch, uint32_t *s_stage) {
    __syncthreads();
    ch.invert_wave(s_stage + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS, true);
    __syncthreads();
}

// "next" row f1: pk[i] = derive_public_key(sk[i])   (ref: goldilocks_ed448_derive_public_key)
// workspace: DERIVE_SLOT_U4 uint4 per operation (xn | yn | denominator | prefix)
template <bool CT>
__device__ __forceinline__ void derive_body(uint8_t *pk, const uint8_t *sk, uint32_t n, const uint4 *table,
                                            uint4 *ws) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[CT ? COMB_BIG_LDS_WORDS : 34 * BLOCK];   // CT: the comb's region too (RestagedCombBig)
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    InvChain ch;
    ch.begin();
    auto first = [&](uint32_t i, bool live, const auto &fb) {
        Ed448DeriveState st;
        const fe zn = ed448_derive_begin(st, sk + 57 * (size_t)i, fb, stage, mk);
        uint4 *slot = ws + (size_t)DERIVE_SLOT_U4 * i;
        if (live) {
            fe_store(slot, st.xn);
            fe_store(slot + 4, st.yn);
        }
        ch.push(slot + 8, zn, live);
    };
    if constexpr (CT) {
        RestagedCombBig fb{s_stage, table};
        for_each_op<CT>(n, [&](uint32_t i, bool live) { first(i, live, fb); });
    } else {
        GlobalBwt bwt_tab{table};
        FixedBwt<GlobalBwt> fb{bwt_tab};
        for_each_op<CT>(n, [&](uint32_t i, bool live) { first(i, live, fb); });
    }
    block_invert_in_stage(ch, s_stage);
    for_each_op_reverse(n, [&](uint32_t i) {
        const uint4 *slot = ws + (size_t)DERIVE_SLOT_U4 * i;
        const fe zi = ch.pop(slot + 8);
        Ed448DeriveState st;
        st.xn = fe_load(slot);
        st.yn = fe_load(slot + 4);
        ed448_derive_finish(pk + 57 * (size_t)i, st, zi);
    });
    // SHAKE256(sk) blocks and the recoded secret scalar do not stay behind in LDS
    lds_wipe_lane(s_stage + threadIdx.x, 34);
    lds_wipe_lane(s_bits + threadIdx.x, 15);
}

// "next" row f1: sig[i] = sign(sk[i], pk[i], msg[i])   (ref: goldilocks_ed448_sign)
// workspace: SIGN_SLOT_U4 uint4 per operation (xn | yn | denominator | prefix | nonce | secret),
// then 64 bytes per lane for the hashed-key seed of the signature in flight
template <bool CT>
__device__ __forceinline__ void sign_body(uint8_t *sig, const uint8_t *sk, const uint8_t *pk, const uint8_t *msgs,
                                          const uint64_t *msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                          const uint8_t *ctx, uint32_t ctx_len, uint32_t n, const uint4 *table,
                                          uint4 *ws) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[CT ? COMB_BIG_LDS_WORDS : 34 * BLOCK];   // CT: the comb's region too (RestagedCombBig)
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    uint8_t *scratch = reinterpret_cast<uint8_t *>(ws + (size_t)SIGN_SLOT_U4 * n) +
                       (size_t)(blockIdx.x * BLOCK + threadIdx.x) * 64;
    // -> false for a message the 32-bit byte counters cannot hold (GOLDILOCKS_AMD_MAX_MESSAGE_BYTES):
    // the lane then signs the empty message and its signature is overwritten with zeros
    auto message = [&](uint32_t i, const uint8_t *&msg, uint32_t &mlen) -> bool {
        msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;
        mlen = fits ? (uint32_t)len64 : 0u;
        return fits;
    };
    InvChain ch;
    ch.begin();
    auto first = [&](uint32_t i, bool live, const auto &fb) {
        const uint8_t *msg;
        uint32_t mlen;
        message(i, msg, mlen);
        Ed448SignState st;
        const fe zn = ed448_sign_begin(st, sk + 57 * (size_t)i, msg, mlen, prehashed, ctx, ctx_len, scratch, fb, stage, mk);
        uint4 *slot = ws + (size_t)SIGN_SLOT_U4 * i;
        if (live) {
            fe_store(slot, st.xn);
            fe_store(slot + 4, st.yn);
            sc_store_u4(slot + 16, st.nonce);
            sc_store_u4(slot + 20, st.secret);
        }
        ch.push(slot + 8, zn, live);
    };
    if constexpr (CT) {
        RestagedCombBig fb{s_stage, table};
        for_each_op<CT>(n, [&](uint32_t i, bool live) { first(i, live, fb); });
    } else {
        GlobalBwt bwt_tab{table};
        FixedBwt<GlobalBwt> fb{bwt_tab};
        for_each_op<CT>(n, [&](uint32_t i, bool live) { first(i, live, fb); });
    }
    block_invert_in_stage(ch, s_stage);
    for_each_op_reverse(n, [&](uint32_t i) {
        uint4 *slot = ws + (size_t)SIGN_SLOT_U4 * i;
        const fe zi = ch.pop(slot + 8);
        Ed448SignState st;
        st.xn = fe_load(slot);
        st.yn = fe_load(slot + 4);
        st.nonce = sc_load_u4(slot + 16);
        st.secret = sc_load_u4(slot + 20);
        const uint8_t *msg;
        uint32_t mlen;
        const bool fits = message(i, msg, mlen);
        ed448_sign_finish(sig + 114 * (size_t)i, st, zi, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx, ctx_len, stage);
        if (!fits)
            for (int k = 0; k < 114; k++) sig[114 * (size_t)i + k] = 0;
        // the nonce and the secret scalar do not stay behind in the workspace
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 16; k < 24; k++) slot[k] = z4;
    });
    // neither does the hashed-key seed (with a message it gives the nonce, with the nonce and a
    // signature the secret scalar), nor the SHAKE256(sk) blocks and recoded scalars in LDS
    {
        uint4 *seed = reinterpret_cast<uint4 *>(scratch);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) seed[k] = z4;
    }
    lds_wipe_lane(s_stage + threadIdx.x, 34);
    lds_wipe_lane(s_bits + threadIdx.x, 15);
}

// "next" row f3: X448.  base == nullptr: derive_public_key through the fixed-base table
// (ref: goldilocks_x448, goldilocks_x448_derive_public_key)
// workspace: X448_SLOT_U4 uint4 per operation (numerator | denominator | prefix)
template <bool CT>
__device__ __forceinline__ void x448_body(uint8_t *shared, int32_t *status, const uint8_t *base,
                                          const uint8_t *scalar, uint32_t n, const uint4 *table, uint4 *ws) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    uint32_t *s_comb = fixed_base_stage<CT>(table);
    InvChain ch;
    ch.begin();
    for_each_op<CT>(n, [&](uint32_t i, bool live) {
        uint32_t w[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(scalar + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        fe num, den;
        if (base) {   // Montgomery ladder: no table, conditional swaps by select
            uint32_t b[14];
            const uint32_t *bs = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
            for (int k = 0; k < 14; k++) b[k] = bs[k];
            sc raw;
#pragma unroll
            for (int k = 0; k < 14; k++) raw.w[k] = w[k];
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, raw);
            x448_ladder(num, den, b, bits);
        } else if constexpr (CT) {
            LdsShuffleCombBig comb{s_comb, threadIdx.x & 63u};
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, comb_big::recode(x448_public_scalar(w)));
            const pt p = ladder_comb(bits, comb);
            num = p.y;
            den = p.x;
        } else {
            GlobalBwt tab{table};
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_bwt(x448_public_scalar(w), tab));
            const pt p = ladder_bwt(bits, tab);
            num = p.y;
            den = p.x;
        }
        uint4 *slot = ws + (size_t)X448_SLOT_U4 * i;
        if (live) fe_store(slot, num);
        ch.push(slot + 4, den, live);
    });
    {   // one inversion per wave: in the comb's region when there is one (nobody needs it any more), else in its own
        __shared__ uint32_t s_inv[CT ? 1 : (BLOCK / 64) * INV_WAVE_LDS_WORDS];
        block_invert_in_stage(ch, CT ? s_comb : s_inv);
    }
    for_each_op_reverse(n, [&](uint32_t i) {
        uint4 *slot = ws + (size_t)X448_SLOT_U4 * i;
        const fe di = ch.pop(slot + 4);
        const fe num = fe_load(slot);
        uint32_t o[14];
        bool ok = true;
        if (base) ok = x448_finish(o, num, di);
        else x448_public_finish(o, num, di);
        uint32_t *dst = reinterpret_cast<uint32_t *>(shared + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = o[k];
        if (status) status[i] = ok ? -1 : 0;
        const uint4 z4 = make_uint4(0, 0, 0, 0);   // the shared secret's numerator does not stay behind
#pragma unroll
        for (int k = 0; k < 4; k++) slot[k] = z4;
    });
    lds_wipe_lane(s_bits + threadIdx.x, 15);   // nor the private scalar in LDS
}

}  // namespace gd
