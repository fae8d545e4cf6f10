// kernels_verify.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "varbase_bodies.hpp"

namespace gd {

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]; b1 == nullptr: b1 is the base point (its 16-bit window table)
GD_KERNEL k_double_scalarmul(uint64_t *out, const uint64_t *__restrict__ b1, const uint64_t *__restrict__ s1,
                             const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                             uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt) {
    double_scalarmul_body(out, b1, s1, b2, s2, n, workspace, bwt);
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
    __shared__ uint32_t s_bits[16 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBitsVerify mk{s_bits + threadIdx.x};
    // Half-size scalars (lattice.hpp): A and R share one ladder of 45 windows; two tables per lane.  The loop is
    // wave-uniform: a lane without a signature of its own in the last round verifies the batch's last one once more
    // and stores nothing.
    __shared__ uint4 s_step[STEP_LDS_U4];   // the table builds' step (LdsStepTable)
    LdsStepTable<> a_tab{lane_table_at(workspace, 0, 2).p, s_step + threadIdx.x},
                   r_tab{lane_table_at(workspace, 1, 2).p, s_step + threadIdx.x};
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t slot = lane + r * stride;
        const bool live = slot < n;
        const uint32_t i = live ? slot : n - 1;
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;   // longer than the 32-bit byte counters hold: the lane fails
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg,
                                                  fits ? (uint32_t)len64 : 0u, prehashed, ctx, ctx_len);
        const bool ok = ed448_verify_lattice(m, b_tab, a_tab, r_tab, stage, mk);
        if (live) status[i] = ok && fits ? -1 : 0;
    }
}

}  // namespace gd
