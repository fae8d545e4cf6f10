// kernels_verify.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "varbase_bodies.hpp"

// 1: half-size scalars, A and R in one ladder of about 46 windows (ed448_verify_lattice; the default).
// 0: the full-length ladder with one exponentiation per signature (ed448_verify_chained): 13 % slower,
//    kept as the measured alternative (profiles/r02/experiments.md).
#ifndef GD_VERIFY_LATTICE
#define GD_VERIFY_LATTICE 1
#endif
#ifndef GD_VERIFY_GROUP   // 1: the signatures of a block are grouped by the length of their pairs (see the kernel)
#define GD_VERIFY_GROUP 1
#endif

namespace gd {

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]; b1 == nullptr: b1 is the base point (shared table)
GD_KERNEL k_double_scalarmul(uint64_t *out, const uint64_t *__restrict__ b1, const uint64_t *__restrict__ s1,
                             const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                             uint4 *__restrict__ workspace, const uint4 *__restrict__ base_tab) {
    double_scalarmul_body<false>(out, b1, s1, b2, s2, n, workspace, base_tab);
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
    // Half-size scalars (lattice.hpp): A and R share one ladder of about 45 windows; two tables per lane.
    LaneTable a_tab = VarTable<false>::at(workspace, 0, 2), r_tab = VarTable<false>::at(workspace, 1, 2);
    const auto wavemax = [](int x) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_xor(x, d, 64);
            x = y > x ? y : x;
        }
        return x;
    };
    const auto message_of = [&](uint32_t j, bool &fits) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[j] : msgs + (size_t)msg_len * j;
        const uint64_t len64 = msg_offsets ? msg_offsets[j + 1] - msg_offsets[j] : (uint64_t)msg_len;
        fits = len64 < MAX_MESSAGE_BYTES;              // longer than the 32-bit byte counters hold: the lane fails
        return ed448_challenge_string(sig + 114 * (size_t)j, pk + 57 * (size_t)j, msg, fits ? (uint32_t)len64 : 0u,
                                      prehashed, ctx, ctx_len);
    };
#if GD_VERIFY_GROUP
    // A wave's ladder is as long as the longest pair of its 64 signatures (45 windows for 94 % of the
    // challenges, 46 for most of the rest).  Every lane prepares its own signature's pair, the block then
    // groups its 256 pairs by window count (a counting sort through LDS), and lane k walks the k-th: the
    // long pairs share a wave, 45.99 -> 45.25 windows per wave on average.
    // LDS of item t: s_bits rows 0..14 / 15..29 = the two scalars; s_stage rows 15..28 = |tau| S, row 29 = flags
    // and length.  s_stage rows 0..14 of a lane's OWN column hold its base-point scalar during the walk.
    __shared__ uint32_t s_perm[BLOCK];
    __shared__ uint32_t s_count[8];
    const uint32_t tid = threadIdx.x;
    LdsMkBits mk_base{s_stage + tid};
    for (uint32_t i0 = blockIdx.x * BLOCK; i0 < n; i0 += stride) {   // whole blocks take every step together
        {
            const uint32_t j = i0 + tid < n ? i0 + tid : n - 1;      // idle lanes redo the last signature
            bool fits;
            const Ed448Msg m = message_of(j, fits);
            const LatticePair pr = ed448_verify_lattice_pair(m, stage);
#pragma unroll
            for (int k = 0; k < 15; k++) {
                s_bits[k * BLOCK + tid] = pr.b1[k];
                s_bits[(15 + k) * BLOCK + tid] = pr.b2[k];
            }
#pragma unroll
            for (int k = 0; k < 14; k++) s_stage[(15 + k) * BLOCK + tid] = pr.ts.w[k];
            s_stage[29 * BLOCK + tid] = (uint32_t)pr.bits << 8 | (fits ? 4u : 0u) | (pr.tau_pos ? 2u : 0u) | (pr.rho_even ? 1u : 0u);
            if (tid < 8) s_count[tid] = 0;
            __syncthreads();
            const int cls = min(max(lattice_windows(pr.bits) - 44, 0), 7);
            const uint32_t rank = atomicAdd(&s_count[cls], 1u);
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (int c = 0; c < 7; c++) before += c < cls ? s_count[c] : 0u;
            s_perm[before + rank] = tid;
            __syncthreads();
        }
        const uint32_t t = s_perm[tid];                              // the item this lane walks
        const uint32_t info = s_stage[29 * BLOCK + t];
        sc ts;
#pragma unroll
        for (int k = 0; k < 14; k++) ts.w[k] = s_stage[(15 + k) * BLOCK + t];
        const int nw = wavemax(lattice_windows((int)(info >> 8)));
        int word;
        uint32_t mask;
        lattice_top_bit(nw, word, mask);
        s_bits[word * BLOCK + t] |= mask;                            // this lane is the only one touching item t now
        s_bits[(15 + word) * BLOCK + t] |= mask;
        const bool live = i0 + t < n;
        const uint32_t j = live ? i0 + t : n - 1;
        bool fits;
        const Ed448Msg m = message_of(j, fits);
        const bool ok = ed448_verify_lattice_walk(m, (info & 2u) != 0, (info & 1u) != 0, ts, LdsBits{s_bits + t},
                                                  LdsBits{s_bits + 15 * BLOCK + t}, nw, b_tab, a_tab, r_tab, mk_base);
        if (live) status[i0 + t] = ok && fits ? -1 : 0;
        __syncthreads();                                             // the next step overwrites every item
    }
#else
    for (uint32_t i0 = blockIdx.x * BLOCK; i0 < n; i0 += stride) {   // whole waves enter the ladder together
        const uint32_t i = i0 + threadIdx.x;
        const bool live = i < n;
        bool fits;
        const Ed448Msg m = message_of(live ? i : n - 1, fits);      // idle lanes redo the last signature
        const bool ok = ed448_verify_lattice(m, b_tab, a_tab, r_tab, stage, mk, wavemax);
        if (live) status[i] = ok && fits ? -1 : 0;
    }
#endif
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
