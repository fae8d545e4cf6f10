// kernels_verify.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "varbase_bodies.hpp"
#include "inv_wave.hpp"

namespace gd {


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
// ctrl: [0] distinct keys seen, [1] keys with a pooled window table, [2] keys with a comb (k_verify_key_mode: at most
// one of the two is non-zero); zeroed by the host
// (seed: drawn per call by the host, so that which keys collide in the set is not a property of the batch alone)
__device__ __forceinline__ uint32_t key_hash(const uint32_t (&w)[15], uint32_t seed) {
    uint32_t h = 0x9e3779b9u ^ seed;
#pragma unroll
    for (int k = 0; k < 15; k++) h = (h ^ w[k]) * 0x85ebca6bu + (h >> 15);
    return h ^ h >> 13;
}
GD_KERNEL k_verify_dedupe(uint32_t *__restrict__ rep, uint32_t *__restrict__ slot_of, uint32_t *__restrict__ key_list,
                          uint32_t *__restrict__ hash_slots, uint32_t hash_mask, uint32_t *__restrict__ ctrl,
                          const uint8_t *__restrict__ pk, uint32_t n, uint32_t seed) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint32_t w[15];
        load_bytes_as_words(w, pk + 57 * (size_t)i, 57, 15);
        uint32_t h = key_hash(w, seed) & hash_mask, owner;
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
// ctrl[1], ctrl[2]: how this batch's keys are served (one thread, after the dedupe)
GD_KERNEL k_verify_key_mode(uint32_t *__restrict__ ctrl, uint32_t n, uint32_t pool_capacity, uint32_t comb_capacity,
                            uint32_t comb_min_per_key, uint32_t wide_min_per_key, uint32_t xwide_min_per_key) {
    if (blockIdx.x || threadIdx.x) return;
    const uint32_t distinct = ctrl[0];
    uint32_t pooled = 0, combed = 0;
    if (distinct <= comb_capacity && (uint64_t)distinct * comb_min_per_key <= n) combed = distinct;
    else if (2 * (uint64_t)distinct <= n && distinct <= pool_capacity) pooled = distinct;
    ctrl[1] = pooled;
    ctrl[2] = combed;
    // teeth per comb: 9 (scalarmul.hpp comb_xwide, 5 combs) for keys that sign a thousand signatures each -- two
    // thousand when the keys are more than 1 024: then their 1 280 entries each are a matter of throughput, not of one
    // lane's latency (tools/probes/wide_comb_probe.py) --, 8 (comb_wide) for hundreds, else 7 (comb_big)
    const uint64_t xwide_from = (uint64_t)xwide_min_per_key * (distinct > 1024u ? 2u : 1u);
    ctrl[3] = !combed ? 0u
              : xwide_min_per_key && (uint64_t)distinct * xwide_from <= n ? (uint32_t)comb_xwide::TEETH
              : wide_min_per_key && (uint64_t)distinct * wide_min_per_key <= n   ? (uint32_t)comb_wide::TEETH
                                                                                 : (uint32_t)comb_big::TEETH;
}
GD_KERNEL k_verify_key_tables(uint4 *__restrict__ pool, uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                              const uint32_t *__restrict__ key_list, const uint8_t *__restrict__ pk) {
    __shared__ uint4 s_step[STEP_LDS_U4];
    const uint32_t pooled = ctrl[1];
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

// ---- a fixed-base COMB per key, when a batch's keys sign many signatures each.
// With a comb of the key the challenge's multiple needs no ladder at all: P = (-h)*A from the key's 4 x 7 x 16 comb
// (15 doublings + 63 mixed additions, scalarmul.hpp comb_big: what the library's own base point gets), S*B added
// from the base point's window table, and P is compared with R as the reference's goldilocks_448_point_eq compares
// it with the decoded R (src/eddsa.c:299-305, src/goldilocks.c:644-653) -- WITHOUT decoding R: the comparison is
// turned into a polynomial identity plus one sign test whose division the lane's signatures share
// (eddsa.hpp ed448_verify_keycomb_begin / _finish).  The equation of src/eddsa.c as it stands, no short pair, no table
// per signature: 150 K multiply-accumulates instead of 522 K.  What it costs is the comb: 432 successive doublings
// and 256 entries per key, 3.5 M multiply-accumulates and, as built here, the time of 14 verifications; so only for
// keys worth it: ctrl[2] != 0 iff the batch averages at least comb_min_per_key signatures per distinct key
// (goldilocks_amd_set_verify_key_combs; 16 by default, 32 below 2^18 signatures) and has no more of them than the
// call's capacity (2^15 by default, KEY_COMBS_MAX at most).
//   k_verify_key_teeth     (kernels_wave.hip) wave k: decode key k, teeth 2^(16 m) * A_k, m < 28, by row arithmetic;
//                          k_verify_key_teeth_lanes: a lane per key instead, when the keys are many
//   k_verify_key_combs     a lane walks a segment of one comb's entries in Gray-code order (one addition of a doubled
//                          tooth per entry) and a wave normalises its segments with one shared inversion (key_combs.hpp)
//   k_verify_key_count / _scan / _scatter   the signatures in the order of their keys (below)
//   k_ed448_verify_keycomb the verification itself, two passes around the lane's shared inversion
// The teeth of MANY keys (more than KEY_TEETH_BY_WAVE_MAX): a lane per key.  One lane's chain of 432 doublings takes
// 1.7 ms however few keys there are, but it costs a sixth of a wave's instructions: 2^15 keys 1.8 ms against 6.4 ms.
GD_KERNEL k_verify_key_teeth_lanes(uint4 *__restrict__ teeth, uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                   const uint32_t *__restrict__ key_list, const uint8_t *__restrict__ pk) {
    const uint32_t combed = ctrl[2];
    if (combed <= (uint32_t)KEY_TEETH_BY_WAVE_MAX) return;
    const int NT = (int)(key_comb_combs(ctrl[3]) * ctrl[3]), spacing = (int)key_comb_spacing(ctrl[3]);
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < combed; k += stride) {
        uint32_t w[15];
        load_bytes_as_words(w, pk + 57 * (size_t)key_list[k], 57, 15);
        pt P;
        key_ok[k] = pt_decode_eddsa_words(P, w) ? 1 : 0;
        uint4 *out = teeth + (size_t)KEY_TEETH_U4 * k;
#pragma unroll 1
        for (int m = 0; m < NT; m++) {          // T_m, and 2 T_m behind the 28 teeth (as k_verify_key_teeth)
            pniels_store(out + 16 * m, pt_to_pniels(P));
            pt_double(P, true);
            pniels_store(out + 16 * (NT + m), pt_to_pniels(P));
            if (m + 1 == NT) break;
#pragma unroll 1
            for (int d = 1; d < spacing; d++) pt_double(P, d + 1 == spacing);
        }
    }
}
// Entry 64 j + idx of a key's comb is T_(6+7j) + sum_{k<6} (+-) T_(k+7j), + iff bit k of idx.  A lane owns a SEGMENT of
// SEG consecutive Gray codes of one comb of one key (the comb's last segment may be shorter): its first entry is the signed
// sum of 7 teeth (6 additions), each further one differs from its predecessor in one sign, i.e. by (+-) 2 T_k (1 addition)
// -- SEG + 5 additions for SEG entries instead of 6 SEG -- and the SEG share one inversion (one per wave, inv_wave.hpp).
// SEG is chosen on the device from the number of keys.  Few keys leave the device idle and the kernel is one wave's
// LATENCY (a lone wave issues an instruction every 7 cycles; tools/combsphases: 17 K cycles per addition, 330 K per
// inversion, 18 K per entry of the second pass), which short segments shorten as long as every segment's wave finds a
// SIMD at once: k_verify_base_part's persistent blocks hold one of a SIMD's two 256-register slots, so the segments are
// as short as leaves at most KEY_COMB_OCC_NUM / KEY_COMB_OCC_DEN = one wave per SIMD (2^10 keys of 5 x 256 entries: 12
// segments of 22 per comb, 960 waves: alone on the device 0.57 ms against 0.66 with 32 and 0.85 with 16, which needs a
// second round on some SIMDs).  Many keys make it a matter of THROUGHPUT, and a wave's inversion is shared by as many
// entries as KEY_COMB_SEG_MAX allows.
// The entries wait unnormalised in their own slots of the comb; chain: 8 uint4 per (key, entry) for the trick.
__host__ __device__ inline uint32_t key_comb_segment(uint32_t combed, uint32_t teeth_per, uint32_t resident_lanes) {
    const uint32_t per_comb = 1u << (teeth_per - 1), combs = combed * key_comb_combs(teeth_per);
    uint64_t room = (uint64_t)resident_lanes * KEY_COMB_OCC_NUM / ((uint64_t)KEY_COMB_OCC_DEN * combs);   // segments per comb
    if (room > per_comb / (uint32_t)KEY_COMB_SEG) room = per_comb / (uint32_t)KEY_COMB_SEG;
    if (room < 1) room = 1;
    const uint32_t seg = (per_comb + (uint32_t)room - 1) / (uint32_t)room;
    return seg > (uint32_t)KEY_COMB_SEG_MAX ? (uint32_t)KEY_COMB_SEG_MAX : seg;
}
GD_KERNEL k_verify_key_combs(uint4 *__restrict__ combs, const uint4 *__restrict__ teeth, const uint32_t *__restrict__ ctrl,
                             uint4 *__restrict__ chain) {
    if (!ctrl[2]) return;                           // (ctrl[3] is 0 then: no geometry to derive)
    const uint32_t teeth_per = ctrl[3], NT = key_comb_combs(teeth_per) * teeth_per, per_comb = 1u << (teeth_per - 1),
                   entries = key_comb_entries(teeth_per);
    const uint32_t combed = ctrl[2], stride = gridDim.x * BLOCK;
    const uint32_t SEG = key_comb_segment(combed, teeth_per, stride), segs = (per_comb + SEG - 1) / SEG;
    const uint32_t per_key = key_comb_combs(teeth_per) * segs, total = combed * per_key;
    // wave-uniform rounds: the segments' shared inversions are ONE inversion per wave (inv_wave.hpp) -- an exponentiation
    // was two thirds of this kernel's instructions (95 K against 21 additions' 51 K per segment of 16)
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    uint32_t *const inv_region = s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS;
    for (uint32_t t0 = blockIdx.x * BLOCK + threadIdx.x; t0 - (threadIdx.x & 63u) < total; t0 += stride) {
        const bool live = t0 < total;
        const uint32_t t = live ? t0 : total - 1;       // (a lane beyond the end repeats the last segment and stores nothing)
        const uint32_t k = t / per_key, j = (t % per_key) / segs, g0 = (t % segs) * SEG;
        const uint32_t cnt = per_comb - g0 < SEG ? per_comb - g0 : SEG;
        const TeethAt tooth{teeth + (size_t)KEY_TEETH_U4 * k}, twice{teeth + (size_t)KEY_TEETH_U4 * k + 16 * NT};
        uint4 *const comb = combs + (size_t)key_comb_u4(teeth_per) * k + 12 * per_comb * j;
        uint4 *const slots = chain + ((size_t)entries * k + per_comb * j) * 8;
        uint32_t idx = g0 ^ (g0 >> 1);
        pt p = pniels_to_pt(tooth.load(teeth_per - 1 + teeth_per * j), false);
#pragma unroll 1
        for (uint32_t b = 0; b + 1 < teeth_per; b++)
            pt_add_pniels(p, tooth.load(b + teeth_per * j), ((idx >> b) & 1u) == 0, true);
        InvChain ch;
        ch.begin();
#pragma unroll 1
        for (uint32_t s = 0;; s++) {
            uint4 *q = comb + 12 * idx;
            // the next step's doubled tooth is requested BEFORE this entry's 20 scattered stores: memory operations return
            // in order, and behind them the load would wait for their acknowledgements
            const bool more = s + 1 < cnt;
            const uint32_t b = more ? (uint32_t)__builtin_ctz(g0 + s + 1) : 0u;     // the Gray bit that flips
            const pniels step = twice.load(b + teeth_per * j);
            gd_keep_order();
            if (live) {
                fe_store(q, fe_weak(fe_sub<2>(p.y, p.x)));
                fe_store(q + 4, fe_weak(fe_add(p.x, p.y)));
                fe_store(q + 8, fe_mulw(p.t, TWO_EFF_D));
            }
            ch.push(slots + 8 * idx, fe_add(p.z, p.z), live);
            if (!more) break;
            idx ^= 1u << b;
            pt_add_pniels(p, step, ((idx >> b) & 1u) == 0, true);
        }
        ch.invert_wave(inv_region, false);
        if (!live) continue;
#pragma unroll 1
        for (uint32_t s = cnt; s-- > 0;) {
            const fe zi = ch.pop(slots + 8 * idx);
            uint4 *q = comb + 12 * idx;
            fe_store(q, fe_mul(fe_load(q), zi));
            fe_store(q + 4, fe_mul(fe_load(q + 4), zi));
            fe_store(q + 8, fe_mul(fe_load(q + 8), zi));
            if (s) idx ^= 1u << (uint32_t)__builtin_ctz(g0 + s);               // back to the predecessor's pattern
        }
    }
}
// The signatures in the order of their keys (a counting sort over the keys' slots): then the lanes of a wave gather
// from ONE key's comb (48 KiB, L2-resident while its signatures are worked on) instead of 64 keys' (2^10 combs are
// 48 MiB: Infinity Cache).  2^20 signatures whose keys arrive in random order: 10.0 -> 9.3 ms when they come sorted.
// Counting and scattering go through per-block bins in LDS, so that a global counter sees one atomic per block and
// key instead of one per signature (16 keys in 2^20 signatures would otherwise queue 65 536 atomics on each of 16
// addresses, twice: + 5 ms).  Beyond KEY_SORT_BINS keys a counter sees few signatures and plain atomics do.
GD_KERNEL k_verify_key_count(uint32_t *__restrict__ count, const uint32_t *__restrict__ rep,
                             const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ ctrl, uint32_t n) {
    __shared__ uint32_t s_bin[KEY_SORT_BINS];
    const uint32_t keys = ctrl[2];
    if (!keys) return;
    const uint32_t stride = gridDim.x * BLOCK;
    if (keys > (uint32_t)KEY_SORT_BINS) {   // many keys: few signatures per counter, plain atomics do
        for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) atomicAdd(count + slot_of[rep[i]], 1u);
        return;
    }
    for (uint32_t k = threadIdx.x; k < keys; k += BLOCK) s_bin[k] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) atomicAdd(s_bin + slot_of[rep[i]], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < keys; k += BLOCK)
        if (s_bin[k]) atomicAdd(count + k, s_bin[k]);
}
// count[k] <- the first position of key k's signatures (one block; exclusive prefix sums over ctrl[2] counts)
GD_KERNEL k_verify_key_scan(uint32_t *__restrict__ count, const uint32_t *__restrict__ ctrl) {
    __shared__ uint32_t s_sum[BLOCK];
    const uint32_t keys = ctrl[2];
    if (!keys || blockIdx.x) return;
    const uint32_t per = (keys + BLOCK - 1) / BLOCK, lo = threadIdx.x * per, hi = lo + per < keys ? lo + per : keys;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += count[k];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < BLOCK; t++) {
            const uint32_t v = s_sum[t];
            s_sum[t] = acc;
            acc += v;
        }
    }
    __syncthreads();
    uint32_t acc = s_sum[threadIdx.x];
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t v = count[k];
        count[k] = acc;
        acc += v;
    }
}
// order[position] = signature: a block counts its signatures per key, reserves that many positions per key with one
// atomic on the key's cursor (count[k], left by the scan), and hands them out from LDS
// (base: what the signature indices written to `order` start at -- `rep` points at the chunk's first signature)
GD_KERNEL k_verify_key_scatter(uint32_t *__restrict__ order, uint32_t *__restrict__ count, const uint32_t *__restrict__ rep,
                               const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ ctrl, uint32_t n,
                               uint32_t base) {
    __shared__ uint32_t s_bin[KEY_SORT_BINS], s_base[KEY_SORT_BINS];
    const uint32_t keys = ctrl[2];
    if (!keys) return;
    const uint32_t stride = gridDim.x * BLOCK;
    if (keys > (uint32_t)KEY_SORT_BINS) {
        for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) order[atomicAdd(count + slot_of[rep[i]], 1u)] = base + i;
        return;
    }
    for (uint32_t k = threadIdx.x; k < keys; k += BLOCK) s_bin[k] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) atomicAdd(s_bin + slot_of[rep[i]], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < keys; k += BLOCK) {
        if (s_bin[k]) s_base[k] = atomicAdd(count + k, s_bin[k]);
        s_bin[k] = 0;
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t k = slot_of[rep[i]];
        order[s_base[k] + atomicAdd(s_bin + k, 1u)] = base + i;
    }
}
// ---- the verification itself, in two kernels around the lanes' shared inversions.
// A lane's positions t (entries of `order`, i.e. signatures in the order of their keys) are verified up to the sign test
// of R's x = L / K by k_ed448_verify_keycomb(_wide) -- the FIRST pass: it parks K, the running product before it, L and
// the flags (KEYCOMB_SLOT_U4 uint4 per position) and leaves the lane's running product in chain_state -- and
// k_ed448_verify_keycomb_finish inverts once per lane and walks the positions back.  Because the running product lives
// in memory between the two, the first pass may come in SEVERAL launches (`resume`): the host-array pipeline verifies a
// batch chunk by chunk as the chunks arrive over PCIe and still shares one inversion between all the positions of a
// lane (goldilocks_amd.hip verify_group); a group of launches covers at most SHARED_INV_OPS_PER_LANE positions per
// resident lane, which bounds the parking space.  All launches of a group use the same grid.
// Positions follow the block's XCD: the hardware deals blocks to the 8 XCDs round robin, so blocks b, b + 8, ... share
// an L2; giving them neighbouring positions -- one key's signatures -- lets one L2 fetch a key's comb instead of four.
__device__ __forceinline__ uint32_t xcd_block() {
    const uint32_t g = gridDim.x;
    return (g & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (g >> 3) + (blockIdx.x >> 3);
}
// S*B of the first q_count positions of a launch, AHEAD of it: S*B needs the signature alone, so it can run -- at one
// block per CU, leaving every CU room for the keys' kernels -- while the combs are being built (the teeth are a chain
// of 432 doublings per key: latency, on a nearly idle device), which takes it out of the verification's first rounds.
// qpark: 16 uint4 per position (a projective niels).
GD_KERNEL k_verify_base_part(uint4 *__restrict__ qpark, const uint8_t *__restrict__ sig, const uint32_t *__restrict__ order,
                             uint32_t q_count, const uint4 *__restrict__ bwt, const uint32_t *__restrict__ ctrl) {
    __shared__ uint32_t s_bits[16 * BLOCK];
    if (!ctrl[2]) return;                           // no combs, no key-comb verification
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsMkBitsVerify mk{s_bits + threadIdx.x};
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < q_count; t += stride)
        pniels_store(qpark + 16 * (size_t)t, ed448_verify_base_part(sig + 114 * (size_t)order[t], b_tab, mk));
}
struct ParkedBase {      // S*B of this position from k_verify_base_part, if its round has it
    const uint4 *slot;
    bool parked;
    __device__ __forceinline__ bool have() const { return parked; }
    __device__ __forceinline__ pniels load() const { return pniels_load(slot); }
};
template <class PLAN>
__device__ __forceinline__ void verify_keycomb_body(const uint8_t *__restrict__ sig,
                                 const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                                 const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                 const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                                 const uint4 *__restrict__ bwt, const uint32_t *__restrict__ rep,
                                 const uint32_t *__restrict__ slot_of, const uint4 *__restrict__ combs,
                                 const uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                 uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                 uint4 *__restrict__ chain_state, uint32_t resume,
                                 const uint4 *__restrict__ qpark, uint32_t q_count, int32_t *__restrict__ finish_status) {
    __shared__ uint32_t s_bits[VERIFY_LDS_WORDS * BLOCK];
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];   // (the inline finish: a region of the wave's own, the block's other waves may still be hashing)
    if (ctrl[3] != (uint32_t)PLAN::TEETH) return;   // this batch's keys are served otherwise (k_ed448_verify, or the other comb)
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsStage stage{s_bits + threadIdx.x};       // (unused by the word-granular absorb)
    LdsMkBitsVerify mk{s_bits + threadIdx.x};
    const uint32_t lane = xcd_block() * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    uint4 *const state = chain_state + 4 * (size_t)lane;
    InvChain ch;
    ch.begin();
    if (resume) ch.acc = fe_load(state);
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; r++) {      // wave-uniform, as in k_ed448_verify; position t: signature i
        const uint32_t pos = lane + r * stride;
        const bool live = pos < n;
        const uint32_t t = live ? pos : n - 1;
        const uint32_t i = order[t];
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg,
                                                  fits ? (uint32_t)len64 : 0u, prehashed, ctx, ctx_len);
        const uint32_t k = slot_of[rep[i]];
        const GlobalCombOf<PLAN> comb{combs + (size_t)PLAN::ENTRIES * 12 * k};
        // whole rounds only (q_count is a multiple of the launch's lanes): uniform for the block
        const ParkedBase q{qpark + 16 * (size_t)t, (r + 1) * stride <= q_count};
        const KeycombPending pend = ed448_verify_keycomb_begin(m, b_tab, comb, stage, mk, q);
        uint4 *slot = park + (size_t)KEYCOMB_SLOT_U4 * t;
        if (live) {
            fe_store(slot + 8, pend.L);
            slot[12] = make_uint4(pend.ok && fits && key_ok[k] != 0 ? 1u : 0u, pend.sign ? 1u : 0u, pend.decided ? 1u : 0u, 0u);
        }
        ch.push(slot, pend.K, live);
    }
    if (!finish_status) {
        fe_store(state, ch.acc);
        return;
    }
    // The group is this one launch (a device-resident batch): the wave inverts and walks back here instead of in
    // k_ed448_verify_keycomb_finish -- the inversion is 42 K dependent instructions per wave, which a kernel of its own
    // runs on an otherwise idle device (0.24 ms per 2^20) and this one beside the waves that are still verifying.
    ch.acc = wave_shared_invert(ch.acc, s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS);
    if (lane >= n) return;
    for (uint32_t pos = lane + (n - 1 - lane) / stride * stride;; pos -= stride) {
        const uint4 *slot = park + (size_t)KEYCOMB_SLOT_U4 * pos;
        const fe inv_k = ch.pop(slot);
        const uint4 flags = slot[12];
        KeycombPending pend;
        pend.L = fe_load(slot + 8);
        pend.ok = flags.x != 0;
        pend.sign = flags.y != 0;
        pend.decided = flags.z != 0;
        finish_status[order[pos]] = ed448_verify_keycomb_finish(pend, inv_k) ? -1 : 0;
        if (pos < stride) break;
    }
}

#define KEYCOMB_ARGS                                                                                                      \
    const uint8_t *__restrict__ sig, const uint8_t *__restrict__ pk,                                                      \
        const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed, \
        const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n, const uint4 *__restrict__ bwt,                     \
        const uint32_t *__restrict__ rep, const uint32_t *__restrict__ slot_of, const uint4 *__restrict__ combs,          \
        const uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl, uint4 *__restrict__ park,                  \
        const uint32_t *__restrict__ order, uint4 *__restrict__ chain_state, uint32_t resume,                            \
        const uint4 *__restrict__ qpark, uint32_t q_count, int32_t *__restrict__ finish_status
GD_KERNEL k_ed448_verify_keycomb(KEYCOMB_ARGS) {        // keys with 7 teeth per comb (4 x 7 x 16)
    verify_keycomb_body<comb_big>(sig, pk, msgs, msg_offsets, msg_len, prehashed, ctx, ctx_len, n, bwt, rep, slot_of, combs,
                                  key_ok, ctrl, park, order, chain_state, resume, qpark, q_count, finish_status);
}
GD_KERNEL k_ed448_verify_keycomb_xwide(KEYCOMB_ARGS) {  // keys with 9 (5 x 9 x 10): a thousand signatures per key
    verify_keycomb_body<comb_xwide>(sig, pk, msgs, msg_offsets, msg_len, prehashed, ctx, ctx_len, n, bwt, rep, slot_of, combs,
                                    key_ok, ctrl, park, order, chain_state, resume, qpark, q_count, finish_status);
}
GD_KERNEL k_ed448_verify_keycomb_wide(KEYCOMB_ARGS) {   // keys with 8 (4 x 8 x 14): hundreds of signatures per key
    verify_keycomb_body<comb_wide>(sig, pk, msgs, msg_offsets, msg_len, prehashed, ctx, ctx_len, n, bwt, rep, slot_of, combs,
                                   key_ok, ctrl, park, order, chain_state, resume, qpark, q_count, finish_status);
}
#undef KEYCOMB_ARGS
// the lanes' inversions and the second pass over every launch of the group, last launch first: chunks.lo[c] is where
// launch c's positions start in `park` / `order` (the pointers its launch was given, relative to the group's)
GD_KERNEL k_ed448_verify_keycomb_finish(int32_t *__restrict__ status, const uint32_t *__restrict__ ctrl,
                                        const uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                        const uint4 *__restrict__ chain_state, VerifyChunks chunks) {
    if (!ctrl[2]) return;                           // no combs: k_ed448_verify has written the verdicts
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    const uint32_t lane = xcd_block() * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    InvChain ch;
    // one inversion per WAVE (inv_wave.hpp), not one per lane: 0.40 -> 0.2x ms for 2^20 signatures
    ch.acc = wave_shared_invert(fe_load(chain_state + 4 * (size_t)lane), s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS);
    for (uint32_t c = chunks.count; c-- > 0;) {
        const uint32_t lo = chunks.lo[c], n = chunks.m[c];
        if (lane >= n) continue;
        for (uint32_t p = lane + (n - 1 - lane) / stride * stride;; p -= stride) {
            const uint32_t t = lo + p, i = order[t];
            const uint4 *slot = park + (size_t)KEYCOMB_SLOT_U4 * t;
            const fe inv_k = ch.pop(slot);
            const uint4 flags = slot[12];
            KeycombPending pend;
            pend.L = fe_load(slot + 8);
            pend.ok = flags.x != 0;
            pend.sign = flags.y != 0;
            pend.decided = flags.z != 0;
            status[i] = ed448_verify_keycomb_finish(pend, inv_k) ? -1 : 0;
            if (p < stride) break;
        }
    }
}

}  // namespace gd
