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

// ---- one decoding and one window table per DISTINCT public key of a batch.
// A verifier's batch usually holds many signatures of few keys (BASELINE config 4: 2^20 signatures of 2^10 keys).
// The key's share of a verification -- its decoding (a 446-squaring exponentiation) and its 16-entry table,
// 15 % of the arithmetic and a third of the table memory -- does not depend on the signature, so it is done once per
// key and the lanes read the key's table from a pool that stays in the caches.  Verdicts are the same lane for lane.
//   k_verify_dedupe      open-addressing hash set of the batch's 57-byte keys (atomicCAS on the index of the first
//                        signature that claims a slot): rep[i] = that signature; representatives take the pool
//                        slots in the order they arrive (ctrl[0] counts them)
//   k_verify_key_tables  decode key k and build its table in pool slot k
//   k_ed448_verify       a lane whose key has a pool slot skips both
// All keys are pooled or none: a batch whose distinct keys do not fit the pool, or in which more than half of the
// signatures bring a key of their own, gains too little (lanes with and without a pooled key in one wave run both
// paths: + 4 % measured with a quarter of the keys pooled), so then no table is pooled (ctrl[1] = 0) and every lane
// decodes its key and builds its table itself, as before.
// ctrl: [0] distinct keys seen, [1] pooled keys (written by k_verify_key_tables' first block), both zeroed by the host
__device__ __forceinline__ uint32_t key_hash(const uint32_t (&w)[15]) {
    uint32_t h = 0x9e3779b9u;
#pragma unroll
    for (int k = 0; k < 15; k++) h = (h ^ w[k]) * 0x85ebca6bu + (h >> 15);
    return h ^ h >> 13;
}
GD_KERNEL k_verify_dedupe(uint32_t *__restrict__ rep, uint32_t *__restrict__ slot_of, uint32_t *__restrict__ key_list,
                          uint32_t *__restrict__ hash_slots, uint32_t hash_mask, uint32_t *__restrict__ ctrl,
                          const uint8_t *__restrict__ pk, uint32_t n) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint32_t w[15];
        load_bytes_as_words(w, pk + 57 * (size_t)i, 57, 15);
        uint32_t h = key_hash(w) & hash_mask, owner;
        for (;;) {
            owner = atomicCAS(hash_slots + h, 0xffffffffu, i);
            if (owner == 0xffffffffu) {
                owner = i;
                break;
            }
            uint32_t v[15];
            load_bytes_as_words(v, pk + 57 * (size_t)owner, 57, 15);
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < 15; k++) diff |= v[k] ^ w[k];
            if (!diff) break;
            h = (h + 1) & hash_mask;
        }
        rep[i] = owner;
        if (owner == i) {
            const uint32_t k = atomicAdd(ctrl, 1u);
            slot_of[i] = k;
            key_list[k] = i;
        }
    }
}
GD_KERNEL k_verify_key_tables(uint4 *__restrict__ pool, uint8_t *__restrict__ key_ok, uint32_t *__restrict__ ctrl,
                              const uint32_t *__restrict__ key_list, const uint8_t *__restrict__ pk, uint32_t n,
                              uint32_t capacity) {
    __shared__ uint4 s_step[STEP_LDS_U4];
    const uint32_t distinct = ctrl[0];
    const uint32_t pooled = 2 * (uint64_t)distinct > n || distinct > capacity ? 0u : distinct;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[1] = pooled;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < pooled; k += stride) {
        uint32_t w[15];
        load_bytes_as_words(w, pk + 57 * (size_t)key_list[k], 57, 15);
        pt A;
        key_ok[k] = pt_decode_eddsa_words(A, w) ? 1 : 0;
        LdsStepTable<> tab{pool + (size_t)KEY_TABLE_U4 * k, s_step + threadIdx.x};
        build_window_table(tab, A);
    }
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
