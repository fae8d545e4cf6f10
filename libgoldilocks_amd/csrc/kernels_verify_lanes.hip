// kernels_verify_lanes.hip -- the kernels that walk TWO digit-addressed window tables per lane: verification when every
// lane decodes its own key (no key of the batch repeats often enough for a comb or a pooled table) and the public
// double-base multiplication.  A translation unit of their own because they are compiled WITHOUT the pair-wise
// additions of gf28.hpp: both live at the register limit (72 - 139 spilled registers), and the aligned register pairs
// cost them more in spills than the additions save (same-box A/B, profiles/r05/ab_pairs.txt: verification of distinct
// keys 33.6 ms without, 34.5 ms with).
#if !defined(GD_PAIRS_EVERYWHERE)   // (A/B builds: tools/build_variants.py)
#define GD_NO_PAIRED_ADDS 1
#endif
#include "varbase_bodies.hpp"

namespace gd {

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]; b1 == nullptr: b1 is the base point (its 16-bit window table)
GD_KERNEL k_double_scalarmul(uint64_t *out, const uint64_t *b1, const uint64_t *__restrict__ s1,
                             const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                             uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt) {
    double_scalarmul_body(out, b1, s1, b2, s2, n, workspace, bwt);
}

// config 4: status[i] = ed448_verify(sig[i], pk[i], msg[i])   (ref: goldilocks_ed448_verify)
GD_KERNEL k_ed448_verify(int32_t *__restrict__ status, const uint8_t *__restrict__ sig,
                         const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                         const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                         const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                         uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt,
                         const uint32_t *__restrict__ rep, const uint32_t *__restrict__ slot_of,
                         const uint4 *__restrict__ pool, const uint8_t *__restrict__ key_ok,
                         const uint32_t *__restrict__ ctrl) {
    __shared__ uint32_t s_bits[16 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    __shared__ uint4 s_step[STEP_LDS_U4];   // the table builds' step (LdsStepTable)
    if (ctrl && ctrl[2]) return;                // this batch's keys have combs (k_ed448_verify_keycomb)
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBitsVerify mk{s_bits + threadIdx.x};
    // Half-size scalars (lattice.hpp): A and R share one ladder of 45 windows; two tables per lane (the key's one
    // unused when the key has a pooled table).  The loop is wave-uniform: a lane without a signature of its own in
    // the last round verifies the batch's last one once more and stores nothing.
    uint4 *const own_a = lane_table_at(workspace, 0, 2).p;
    LdsStepTable<> r_tab{lane_table_at(workspace, 1, 2).p, s_step + threadIdx.x};
    const uint32_t pooled = ctrl ? ctrl[1] : 0u;
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
        uint32_t k = 0xffffffffu;
        if (pooled) k = slot_of[rep[i]];
        const bool shared = k < pooled;
        LdsStepTable<> a_tab{shared ? const_cast<uint4 *>(pool) + (size_t)KEY_TABLE_U4 * k : own_a, s_step + threadIdx.x};
        const bool ok = ed448_verify_lattice(m, b_tab, a_tab, r_tab, stage, mk, shared, shared ? key_ok[k] != 0 : true);
        if (live) status[i] = ok && fits ? -1 : 0;
    }
}

}  // namespace gd
