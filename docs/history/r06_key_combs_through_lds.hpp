// RECORD of a variant that was built, parity-green, measured and NOT adopted (docs/history/r06.md; tools/combsphases times
// it as ORDER 3): k_verify_key_combs with its records moved through LDS.  2^10 keys, alone on the device: 0.86 ms against
// the product's 0.57 (59 - 69 spilled registers; a reload from scratch waits behind the scattered stores like any load).
//
// key_combs.hpp -- the entries of the per-key combs of large verification batches (kernels_verify.hip k_verify_key_combs;
// what the combs are for: eddsa.hpp ed448_verify_keycomb_begin, reference src/eddsa.c:253-306 with (-h)*A from a comb).
//
// Entry 64 j + idx of a key's comb is T_(6+7j) + sum_{k<6} (+-) T_(k+7j), + iff bit k of idx (7 teeth per comb; 8 and 9
// likewise).  A lane owns a SEGMENT of SEG consecutive Gray codes of one comb of one key (the comb's last segment may be
// shorter): its first entry is the signed sum of the comb's teeth, each further one differs from its predecessor in one
// sign, i.e. by (+-) 2 T_k (1 addition), and the segments of a wave share one inversion (inv_wave.hpp) to normalise
// their entries: pass 1 walks the segment and leaves the raw entries in their places of the comb, pass 2 walks back and
// multiplies them by 1 / 2Z.
//
// SEG is chosen on the device from the number of keys (key_comb_segment).  Few keys leave the device idle and the kernel
// is one wave's LATENCY (a lone wave issues an instruction every 7 cycles, tools/combsphases: 17 K cycles per addition,
// 330 K per inversion), which short segments shorten as long as every segment's wave finds a SIMD at once:
// k_verify_base_part's persistent blocks hold one of a SIMD's two 256-register slots, so the segments are as short as
// leaves at most KEY_COMB_OCC_NUM / KEY_COMB_OCC_DEN = one wave per SIMD (2^10 keys of 5 x 256 entries: 12 segments of 22
// per comb, 960 waves).  Many keys make it a matter of THROUGHPUT, and a wave's inversion is shared by as many entries
// as KEY_COMB_SEG_MAX allows.
//
// MEMORY.  A lane's entries are 192-byte records far from its neighbours' (a segment apart): moved lane by lane, every
// 16-byte access is a cache line of its own, 64 lines per instruction, and the CU's one address pipeline -- a line per
// cycle -- was what pass 2 waited for (32 such instructions per entry and wave, four waves per CU: 8 K of its 20 K cycles
// per entry; tools/combsphases).  So (a) the records go through LDS (WaveEntries): the wave's 64 records are read and
// written by instructions whose consecutive lanes take consecutive 16 bytes of a record, 5.3 records = 11 - 13 lines per
// instruction; (b) what only this kernel reads -- the chain of the shared inversion, 128 bytes per entry -- lies
// lane-interleaved in a region of the wave's own (8 lines per instruction); (c) nothing is requested right behind a
// store that it would have to wait for (memory operations return in order): pass 1 asks for the next step's doubled
// tooth before it stores, pass 2 for its entry and its chain slot before the chain's two multiplications.
#pragma once
#include "fixed_bodies.hpp"
#include "inv_wave.hpp"

namespace gd {

__host__ __device__ inline uint32_t key_comb_segment(uint32_t combed, uint32_t teeth_per, uint32_t resident_lanes) {
    const uint32_t per_comb = 1u << (teeth_per - 1), combs = combed * key_comb_combs(teeth_per);
    uint64_t room = (uint64_t)resident_lanes * KEY_COMB_OCC_NUM / ((uint64_t)KEY_COMB_OCC_DEN * combs);   // segments per comb
    if (room > per_comb / (uint32_t)KEY_COMB_SEG) room = per_comb / (uint32_t)KEY_COMB_SEG;
    if (room < 1) room = 1;
    const uint32_t seg = (per_comb + (uint32_t)room - 1) / (uint32_t)room;
    return seg > (uint32_t)KEY_COMB_SEG_MAX ? (uint32_t)KEY_COMB_SEG_MAX : seg;
}
// uint4 of chain space that `entries` comb entries need at most: 8 per entry of every segment's full length (a comb's
// last segment may be up to an eighth of the comb short of it) and the last wave's 64 lanes
__host__ __device__ constexpr size_t key_comb_chain_u4(size_t entries) { return (entries + entries / 8 + 64 * (size_t)KEY_COMB_SEG_MAX) * 8; }

// A wave's 64 records of 12 uint4 (192 contiguous bytes each, anywhere in memory) between the lanes' registers and memory
// with coalesced instructions.  Every lane of the wave must call; a lane without a record passes nullptr.
constexpr int WAVE_ENTRIES_ROW_U4 = 13;                                     // (the lanes' rows spread over the banks)
constexpr int WAVE_ENTRIES_LDS_U4 = 64 * WAVE_ENTRIES_ROW_U4 + 64 / 2;      // per wave: the rows, then 64 addresses
struct WaveEntries {
    uint4 *rows;
    struct InFlight { uint4 v[12]; };
    __device__ __forceinline__ uint64_t *addresses() const { return reinterpret_cast<uint64_t *>(rows + 64 * WAVE_ENTRIES_ROW_U4); }
    __device__ __forceinline__ static uint32_t lane() { return threadIdx.x & 63u; }
    __device__ __forceinline__ void store(uint4 *q, const fe &a, const fe &b, const fe &c) const {
        const uint32_t l = lane();
        wave_sync();                                       // (the previous use of the rows has been read)
        fe_store(rows + WAVE_ENTRIES_ROW_U4 * l, a);
        fe_store(rows + WAVE_ENTRIES_ROW_U4 * l + 4, b);
        fe_store(rows + WAVE_ENTRIES_ROW_U4 * l + 8, c);
        addresses()[l] = reinterpret_cast<uint64_t>(q);
        wave_sync();
#pragma unroll
        for (uint32_t i = 0; i < 12; i++) {
            const uint32_t m = 64 * i + l, e = m / 12u, p = m - 12u * e;
            uint4 *dst = reinterpret_cast<uint4 *>(addresses()[e]);
            const uint4 v = rows[WAVE_ENTRIES_ROW_U4 * e + p];
            if (dst) dst[p] = v;
        }
    }
    // the 12 loads of this lane's share of the wave's records, issued; finish() waits for them
    __device__ __forceinline__ InFlight begin_load(const uint4 *q) const {
        const uint32_t l = lane();
        wave_sync();
        addresses()[l] = reinterpret_cast<uint64_t>(q);
        wave_sync();
        InFlight f;
#pragma unroll
        for (uint32_t i = 0; i < 12; i++) {
            const uint32_t m = 64 * i + l, e = m / 12u, p = m - 12u * e;
            const uint4 *src = reinterpret_cast<const uint4 *>(addresses()[e]);
            f.v[i] = src ? src[p] : make_uint4(0, 0, 0, 0);
        }
        return f;
    }
    __device__ __forceinline__ void finish_load(const InFlight &f, fe &a, fe &b, fe &c) const {
        const uint32_t l = lane();
#pragma unroll
        for (uint32_t i = 0; i < 12; i++) {
            const uint32_t m = 64 * i + l, e = m / 12u, p = m - 12u * e;
            rows[WAVE_ENTRIES_ROW_U4 * e + p] = f.v[i];
        }
        wave_sync();
        a = fe_load(rows + WAVE_ENTRIES_ROW_U4 * l);
        b = fe_load(rows + WAVE_ENTRIES_ROW_U4 * l + 4);
        c = fe_load(rows + WAVE_ENTRIES_ROW_U4 * l + 8);
    }
};
// a field element in four pieces 64 uint4 apart (a wave's lane-interleaved rows)
__device__ __forceinline__ fe fe_load_wave_rows(const uint4 *p) { return fe_from_u4(p[0], p[64], p[128], p[192]); }
__device__ __forceinline__ void fe_store_wave_rows(uint4 *p, const fe &a) {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[64] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    p[128] = make_uint4(a.v[8], a.v[9], a.v[10], a.v[11]);
    p[192] = make_uint4(a.v[12], a.v[13], a.v[14], a.v[15]);
}

struct NoClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
};
// combs: key k's comb at key_comb_u4(teeth_per) * k, entry (j, idx) 12 uint4 (niels: y - x, x + y, 2 d t, all over 2 z)
// teeth: key k's NT teeth and behind them their doubles, pniels of 16 uint4 (k_verify_key_teeth)
// chain: key_comb_chain_u4(entries of all keys) uint4 of scratch;  lds: WAVE_ENTRIES_LDS_U4 uint4 per wave of the block
// CLOCK: tools/combsphases reads the clock between the phases
template <class CLOCK>
__device__ __forceinline__ void verify_key_combs_body(uint4 *__restrict__ combs, const uint4 *__restrict__ teeth, uint32_t combed,
                                                      uint32_t teeth_per, uint32_t SEG, uint4 *__restrict__ chain, uint4 *lds,
                                                      CLOCK &clock) {
    const uint32_t NT = key_comb_combs(teeth_per) * teeth_per, per_comb = 1u << (teeth_per - 1);
    const uint32_t stride = gridDim.x * BLOCK, segs = (per_comb + SEG - 1) / SEG;
    const uint32_t per_key = key_comb_combs(teeth_per) * segs, total = combed * per_key, l = threadIdx.x & 63u;
    const WaveEntries entries{lds + (threadIdx.x >> 6) * WAVE_ENTRIES_LDS_U4};
    static_assert(WAVE_ENTRIES_LDS_U4 * 4 >= INV_WAVE_LDS_WORDS, "the wave's inversion borrows its rows");
    uint32_t *const inv_region = reinterpret_cast<uint32_t *>(entries.rows);
    // wave-uniform rounds and steps: the LDS transposition and the shared inversion take the whole wave
    for (uint32_t t0 = blockIdx.x * BLOCK + threadIdx.x; t0 - l < total; t0 += stride) {
        const bool live = t0 < total;
        const uint32_t t = live ? t0 : total - 1;       // (a lane beyond the end repeats the last segment and stores nothing)
        const uint32_t k = t / per_key, j = (t % per_key) / segs, g0 = (t % segs) * SEG;
        const uint32_t cnt = per_comb - g0 < SEG ? per_comb - g0 : SEG;
        const TeethAt tooth{teeth + (size_t)KEY_TEETH_U4 * k}, twice{teeth + (size_t)KEY_TEETH_U4 * k + 16 * NT};
        uint4 *const comb = combs + (size_t)key_comb_u4(teeth_per) * k + 12 * per_comb * j;
        uint4 *const slots = chain + (size_t)((t0 - l) >> 6) * SEG * 512 + l;     // step s: 8 rows of 64 uint4 from 512 s on
        uint32_t idx = g0 ^ (g0 >> 1);
        clock.start();
        pt p = pniels_to_pt(tooth.load(teeth_per - 1 + teeth_per * j), false);
#pragma unroll 1
        for (uint32_t b = 0; b + 1 < teeth_per; b++)
            pt_add_pniels(p, tooth.load(b + teeth_per * j), ((idx >> b) & 1u) == 0, true);
        clock.mark(0);
        InvChain ch;
        ch.begin();
#pragma unroll 1
        for (uint32_t s = 0;; s++) {
            const bool act = live && s < cnt, more = s + 1 < cnt;
            const uint32_t b = more ? (uint32_t)__builtin_ctz(g0 + s + 1) : 0u;     // the Gray bit that flips
            const pniels step = twice.load(b + teeth_per * j);                       // (ahead of the stores)
            gd_keep_order();
            entries.store(act ? comb + 12 * idx : nullptr, fe_weak(fe_sub<2>(p.y, p.x)), fe_weak(fe_add(p.x, p.y)), fe_mulw(p.t, TWO_EFF_D));
            {   // InvChain::push into the wave's rows
                const fe z = fe_add(p.z, p.z);
                const bool zero = fe_is_zero(z);
                const fe ze = fe_select(fe_weak(z), fe_one(), zero);
                fe_store_wave_rows(slots + 512 * s, ze);
                fe_store_wave_rows(slots + 512 * s + 256, fe_select(ch.acc, fe_zero(), zero));
                ch.acc = fe_select(ch.acc, fe_mul(ch.acc, ze), act);
            }
            if (s + 1 == SEG) break;
            if (more) idx ^= 1u << b;
            pt_add_pniels(p, step, ((idx >> b) & 1u) == 0, true);
        }
        clock.mark(1);
        ch.invert_wave(inv_region, false);
        clock.mark(2);
        // walking back: a step's entry and its chain slot are requested before the previous step's products are stored
        // (nothing waits behind a store), and the chain's two multiplications stand between the request and the use
#pragma unroll 1
        for (uint32_t s = SEG; s-- > 0;) {
            const bool act = live && s < cnt;
            uint4 *const q = act ? comb + 12 * idx : nullptr;
            const WaveEntries::InFlight raw = entries.begin_load(q);
            const fe ze = fe_load_wave_rows(slots + 512 * s), pre = fe_load_wave_rows(slots + 512 * s + 256);
            gd_keep_order();
            const fe zi = fe_mul(ch.acc, pre);
            ch.acc = fe_select(ch.acc, fe_mul(ch.acc, ze), act);
            fe a, b, cn;
            entries.finish_load(raw, a, b, cn);
            entries.store(q, fe_mul(a, zi), fe_mul(b, zi), fe_mul(cn, zi));
            if (act && s) idx ^= 1u << (uint32_t)__builtin_ctz(g0 + s);               // back to the predecessor's pattern
        }
        clock.mark(3);
    }
}

}  // namespace gd
