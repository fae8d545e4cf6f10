// kernels_verify.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "varbase_bodies.hpp"

// 1: half-size scalars, A and R in one ladder of 45 windows (ed448_verify_lattice; the default).
// 0: the full-length ladder with one exponentiation per signature (ed448_verify_chained): 13 % slower,
//    kept as the measured alternative (profiles/r02/experiments.md).
#ifndef GD_VERIFY_LATTICE
#define GD_VERIFY_LATTICE 1
#endif

namespace gd {

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]; b1 == nullptr: b1 is the base point (its 16-bit window table)
GD_KERNEL k_double_scalarmul(uint64_t *out, const uint64_t *__restrict__ b1, const uint64_t *__restrict__ s1,
                             const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                             uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt) {
    double_scalarmul_body<false>(out, b1, s1, b2, s2, n, workspace, bwt);
}

// test hook: the short pair (rho, tau) of verification's half-size scalars for challenge h[i] (lattice.hpp)
GD_KERNEL k_half_size_pair(uint32_t *__restrict__ rho, uint32_t *__restrict__ tau, const uint64_t *__restrict__ h,
                           uint32_t n) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        wide15 r;
        int8w t;
        half_size_pair(r, t, sc_load_abi(h + 7 * (size_t)i));
#pragma unroll
        for (int k = 0; k < 15; k++) rho[15 * (size_t)i + k] = r.w[k];
#pragma unroll
        for (int k = 0; k < 8; k++) tau[8 * (size_t)i + k] = t.w[k];
    }
}

// config 4: status[i] = ed448_verify(sig[i], pk[i], msg[i])   (ref: goldilocks_ed448_verify)
GD_KERNEL k_ed448_verify(int32_t *__restrict__ status, const uint8_t *__restrict__ sig,
                         const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                         const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                         const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                         uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_bits[30 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
#if GD_VERIFY_LATTICE
    // Half-size scalars (lattice.hpp): A and R share one ladder of 45 windows; two tables per lane.
    LaneTable a_tab = VarTable<false>::at(workspace, 0, 2), r_tab = VarTable<false>::at(workspace, 1, 2);
    for (uint32_t i = lane; i < n; i += stride) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;   // longer than the 32-bit byte counters hold: the lane fails
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg,
                                                  fits ? (uint32_t)len64 : 0u, prehashed, ctx, ctx_len);
        const bool ok = ed448_verify_lattice(m, b_tab, a_tab, r_tab, stage, mk);
        status[i] = ok && fits ? -1 : 0;
    }
#else
    LaneTable a_tab{workspace + (size_t)lane * TABLE_U4};
    // One exponentiation per signature: each verification hands a pending quotient to the next one this
    // lane handles (ed448_verify_chained, eddsa.hpp).
    VerifyPending pend;
    verify_pending_clear(pend);
    for (uint32_t i = lane; i < n; i += stride) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;   // longer than the 32-bit byte counters hold: the lane fails
        const uint32_t mlen = fits ? (uint32_t)len64 : 0u;
        Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx,
                                            ctx_len);
        uint32_t done_index[2];
        bool done_ok[2];
        const int nd = ed448_verify_chained(m, i, pend, b_tab, a_tab, stage, mk, done_index, done_ok);
        if (!fits) pend.ok = false;
        for (int k = 0; k < nd; k++) status[done_index[k]] = done_ok[k] && (done_index[k] != i || fits) ? -1 : 0;
    }
    if (pend.live) {
        uint32_t idx;
        const bool v = ed448_verify_chain_flush(pend, idx);
        status[idx] = v ? -1 : 0;
    }
#endif
}

}  // namespace gd
