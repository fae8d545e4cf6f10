// kernels.hpp -- HIP kernels for gfx950: one Ed448 operation per wavefront lane.
//
// Memory plan:
//   * I/O arrays are the reference's AoS structs (256-B points, 56-B scalars); a wave moves the
//     contiguous block of its 64 operations with fully coalesced 16-byte-per-lane instructions and
//     transposes it through LDS (wave_load_points / wave_store_points below).  I/O is ~568 B per
//     ~4000 field multiplications: not the bound.
//   * recoded scalars live in LDS, word-major ([word][lane]): window/comb bit positions
//     are wave-uniform, so every LDS read is conflict-free.
//   * the per-lane window table of a variable base (16 projective niels = 4 KiB) cannot
//     fit LDS (64 lanes x 4 KiB = 256 KiB per wave), so it lives in an HBM workspace,
//     lane-contiguous, sized by RESIDENT lanes (grid-stride), read 256 B per window.
//   * shared read-only tables: a comb (15 KiB) is staged in LDS and gathered with wavefront
//     shuffles; the base point's 16-entry window table (4 KiB) and its 28 x 32768 16-bit window table
//     (168 MiB, Infinity-Cache resident) are global buffers.
//   * kernels that end in a field inversion park per-operation state in the workspace and share
//     one inversion between the operations of a lane (fixed_bodies.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "abi.hpp"
#include "eddsa.hpp"
#include "scalarmul.hpp"
#include "x448.hpp"

namespace gd {

constexpr int BLOCK = 256;          // 4 waves: one per SIMD
#ifndef GD_WAVES_PER_SIMD
#define GD_WAVES_PER_SIMD 2
#endif
constexpr int WAVES_PER_SIMD = GD_WAVES_PER_SIMD;   // 2 blocks per CU -> 256-VGPR budget per lane
constexpr int TABLE_U4 = 17 * 16;    // uint4 per lane window table (16 entries + 1 build slot, x 4 fe x 4 uint4)
constexpr int KEY_TABLE_U4 = 16 * 16;   // uint4 per pooled key table of verification: 16 entries of 256 bytes
constexpr int PRECOMP_U4 = 80 * 20 + 64;  // uint4 per lane for k_precompute: 80 x 5 fe + 4 doubled teeth
// per-OPERATION workspace of the kernels that share one inversion between a lane's operations
// (fixed_bodies.hpp): numerator(s) | denominator | prefix product [| nonce | secret scalar]
constexpr int DERIVE_SLOT_U4 = 16, SIGN_SLOT_U4 = 24, X448_SLOT_U4 = 12;
constexpr int ML_SLOT_U4 = 8;   // the table-free variable-base ladder: denominator | prefix product
// the two-ladder kernels park the FIRST product of an operation in its slots (free once the chain has been popped) until
// every input of the operation has been read: any output may then alias any input, as in the reference
constexpr int ML_DUAL_SLOT_U4 = 16;     // point_dual_scalarmul: chain slot, later s2 * P (one point = 16 uint4)
constexpr int ML_DOUBLE_SLOT_U4 = 76;   // point_double_scalarmul: four chain slots | four numerators | p1 | a, b, c (varbase_bodies.hpp)
constexpr int SHARED_INV_OPS_PER_LANE = 8;   // a launch covers at most this many operations per resident lane
constexpr uint64_t MAX_MESSAGE_BYTES = 0x7fffff00ull;   // GOLDILOCKS_AMD_MAX_MESSAGE_BYTES: byte counters are 32-bit

// ---------------------------------------------------------------- register <-> memory

__device__ __forceinline__ fe fe_from_u4(const uint4 &q0, const uint4 &q1, const uint4 &q2, const uint4 &q3) {
    fe r;
    r.v[0] = q0.x; r.v[1] = q0.y; r.v[2] = q0.z; r.v[3] = q0.w;
    r.v[4] = q1.x; r.v[5] = q1.y; r.v[6] = q1.z; r.v[7] = q1.w;
    r.v[8] = q2.x; r.v[9] = q2.y; r.v[10] = q2.z; r.v[11] = q2.w;
    r.v[12] = q3.x; r.v[13] = q3.y; r.v[14] = q3.z; r.v[15] = q3.w;
    return r;
}
__device__ __forceinline__ fe fe_load(const uint4 *p) { return fe_from_u4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void fe_store(uint4 *p, const fe &a) {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    p[2] = make_uint4(a.v[8], a.v[9], a.v[10], a.v[11]);
    p[3] = make_uint4(a.v[12], a.v[13], a.v[14], a.v[15]);
}

// gf_448_s (8 x u64) at 16-byte aligned address
__device__ __forceinline__ fe fe_load_abi(const uint64_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    uint64_t l[8] = {(uint64_t)a.x | (uint64_t)a.y << 32, (uint64_t)a.z | (uint64_t)a.w << 32,
                     (uint64_t)b.x | (uint64_t)b.y << 32, (uint64_t)b.z | (uint64_t)b.w << 32,
                     (uint64_t)c.x | (uint64_t)c.y << 32, (uint64_t)c.z | (uint64_t)c.w << 32,
                     (uint64_t)d.x | (uint64_t)d.y << 32, (uint64_t)d.z | (uint64_t)d.w << 32};
    return fe_weak(fe_from_limbs56(l));
}
__device__ __forceinline__ void fe_store_abi(uint64_t *p, const fe &a) {
    uint64_t l[8];
    fe_to_limbs56(l, a);
    uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++)
        q[i] = make_uint4((uint32_t)l[2 * i], (uint32_t)(l[2 * i] >> 32), (uint32_t)l[2 * i + 1],
                          (uint32_t)(l[2 * i + 1] >> 32));
}
__device__ __forceinline__ pt pt_load_abi(const uint64_t *p) {
    pt r;
    r.x = fe_load_abi(p);
    r.y = fe_load_abi(p + 8);
    r.z = fe_load_abi(p + 16);
    r.t = fe_load_abi(p + 24);
    return r;
}
__device__ __forceinline__ void pt_store_abi(uint64_t *p, const pt &r) {
    fe_store_abi(p, r.x);
    fe_store_abi(p + 8, r.y);
    fe_store_abi(p + 16, r.z);
    fe_store_abi(p + 24, r.t);
}
// scalar_s: 7 x u64, only 8-byte aligned
__device__ __forceinline__ sc sc_load_abi(const uint64_t *p) {
    uint64_t l[7];
#pragma unroll
    for (int i = 0; i < 7; i++) l[i] = p[i];
    return sc_from_abi(l);
}

// ---------------------------------------------------------------- wave-cooperative coalesced I/O
// The reference's arrays are arrays of structs.  A lane that fetched its own 256-byte point directly
// would issue 16-byte accesses 256 bytes apart from its neighbours': every byte of every line is
// used, but each instruction touches 64 lines.  The headline kernels instead move a wave's
// contiguous block (64 points = 16 KiB, 64 scalars = 3.5 KiB) with fully coalesced instructions --
// lane l takes the l-th 16 (points) or 8 (scalars, only 8-byte aligned) bytes of every row -- and
// transpose through a per-wave LDS buffer, 32 points at a time.  i0 = the wave's first operation,
// m = how many of its 64 operations exist (both wave-uniform); every lane of the wave must call.
constexpr int WAVE_STAGE_U4 = 512;   // 8 KiB of LDS per wave
__device__ __forceinline__ void wave_sync() {   // LDS writes of this wave before, LDS reads after
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ pt wave_load_points(uint4 *stage, const uint64_t *base, uint32_t i0, uint32_t m, uint32_t l) {
    const uint4 *g = reinterpret_cast<const uint4 *>(base) + (size_t)i0 * 16;
    pt r = pt_identity();
#pragma unroll 1
    for (uint32_t half = 0; half < 2; half++) {
        wave_sync();
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t idx = half * 512 + k * 64 + l;
            if (idx < m * 16) stage[k * 64 + l] = g[idx];
        }
        wave_sync();
        if ((l >> 5) == half && l < m) r = pt_load_abi(reinterpret_cast<const uint64_t *>(stage + (l & 31) * 16));
    }
    return r;
}
__device__ __forceinline__ void wave_store_points(uint4 *stage, uint64_t *out, uint32_t i0, uint32_t m, uint32_t l,
                                                  const pt &r) {
    uint4 *g = reinterpret_cast<uint4 *>(out) + (size_t)i0 * 16;
#pragma unroll 1
    for (uint32_t half = 0; half < 2; half++) {
        wave_sync();
        if ((l >> 5) == half && l < m) pt_store_abi(reinterpret_cast<uint64_t *>(stage + (l & 31) * 16), r);
        wave_sync();
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t idx = half * 512 + k * 64 + l;
            if (idx < m * 16) g[idx] = stage[k * 64 + l];
        }
    }
}
__device__ __forceinline__ sc wave_load_scalars(uint4 *stage, const uint64_t *scalar, uint32_t i0, uint32_t m, uint32_t l) {
    const uint64_t *g = scalar + (size_t)i0 * 7;
    uint64_t *s64 = reinterpret_cast<uint64_t *>(stage);
    wave_sync();
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) {
        const uint32_t idx = k * 64 + l;
        if (idx < m * 7) s64[idx] = g[idx];
    }
    wave_sync();
    return l < m ? sc_load_abi(s64 + l * 7) : sc_zero();
}

// ---------------------------------------------------------------- policies

struct LdsBits {  // recoded scalar, word-major in LDS
    const uint32_t *p;
    __device__ __forceinline__ uint32_t word(int k) const { return p[k * BLOCK]; }
};
__device__ __forceinline__ LdsBits lds_put_bits(uint32_t *slot, const sc &s) {
#pragma unroll
    for (int k = 0; k < 14; k++) slot[k * BLOCK] = s.w[k];
    slot[14 * BLOCK] = 0;
    return LdsBits{slot};
}
// Secret scalars do not stay behind in LDS when a kernel ends (the reference zeroizes its
// temporaries, src/goldilocks.c:461-464): every lane clears its own slot(s).
__device__ __forceinline__ void lds_wipe_lane(uint32_t *slot, int words) {
#pragma unroll 1
    for (int k = 0; k < words; k++) slot[k * BLOCK] = 0;
}

__device__ __forceinline__ void pt_store_u4(uint4 *q, const pt &p) {   // a point parked in workspace: 16 uint4
    fe_store(q, p.x);
    fe_store(q + 4, p.y);
    fe_store(q + 8, p.z);
    fe_store(q + 12, p.t);
}
__device__ __forceinline__ pt pt_load_u4(const uint4 *q) {
    pt p;
    p.x = fe_load(q);
    p.y = fe_load(q + 4);
    p.z = fe_load(q + 8);
    p.t = fe_load(q + 12);
    return p;
}
__device__ __forceinline__ void pniels_store(uint4 *q, const pniels &e) {
    fe_store(q, e.a);
    fe_store(q + 4, e.b);
    fe_store(q + 8, e.cn);
    fe_store(q + 12, e.z);
}
__device__ __forceinline__ pniels pniels_load(const uint4 *q) {
    pniels e;
    e.a = fe_load(q);
    e.b = fe_load(q + 4);
    e.cn = fe_load(q + 8);
    e.z = fe_load(q + 12);
    return e;
}
__device__ __forceinline__ void niels_store(uint4 *q, const niels &e) {   // 12 uint4: an affine entry of a comb
    fe_store(q, e.a);
    fe_store(q + 4, e.b);
    fe_store(q + 8, e.cn);
}
// The 28 teeth of a 4 x 7 x 16 comb as pniels, 16 uint4 each, in LDS or in global memory (scalarmul.hpp comb_big_entry)
struct TeethAt {
    const uint4 *p;
    __device__ __forceinline__ pniels load(uint32_t m) const { return pniels_load(p + 16 * m); }
};
using LdsTeeth = TeethAt;
struct LaneTable {  // this lane's window table in the HBM workspace, lane-contiguous; the digit picks the address
    static constexpr bool direct = true;   // lookup = one entry's loads (a two-table ladder may issue both entries' at once)
    uint4 *p;
    __device__ __forceinline__ void store(int k, const pniels &e) const { pniels_store(p + 16 * k, e); }
    __device__ __forceinline__ pniels load(uint32_t k) const { return pniels_load(p + 16 * k); }
    __device__ __forceinline__ pniels lookup(uint32_t idx) const { return load(idx); }
    __device__ __forceinline__ void put_step(const pniels &e) const { store(16, e); }   // slot 16: the build slot
    __device__ __forceinline__ pniels step() const { return load(16); }
};
// The same table with the build's step in LDS instead of slot 16.  Memory operations complete in order, so a
// read issued behind the 16 stores of an entry waits until those have been acknowledged: with the step in the
// table's own memory every iteration of the build pays that (tools/verifyphases: the table builds of
// verification ran at 14-18 clocks per multiply-accumulate against the ladder's 10, and at the ladder's rate
// with the stores left out).  From LDS the step comes back at once and the stores drain behind the next addition.
// Piece-major ([piece][lane], 16 bytes each): conflict-free.  STRIDE = lanes per piece row: 256 for a region of
// the block (+ threadIdx.x), 64 for a region of the wave (+ lane in the wave), which a kernel may reuse for its
// wave's I/O staging between operations.
constexpr int STEP_LDS_U4 = 16 * 256;   // uint4 per 256-lane block
template <int STRIDE = 256>
struct LdsStepTable {
    static constexpr bool direct = true;
    uint4 *p;        // this lane's table in the workspace
    uint4 *lds;      // the step region + this lane's position in a piece row
    __device__ __forceinline__ void store(int k, const pniels &e) const { pniels_store(p + 16 * k, e); }
    __device__ __forceinline__ pniels load(uint32_t k) const { return pniels_load(p + 16 * k); }
    __device__ __forceinline__ pniels lookup(uint32_t idx) const { return load(idx); }
    __device__ __forceinline__ void put_piece(int i, const fe &f) const {
        lds[(4 * i + 0) * STRIDE] = make_uint4(f.v[0], f.v[1], f.v[2], f.v[3]);
        lds[(4 * i + 1) * STRIDE] = make_uint4(f.v[4], f.v[5], f.v[6], f.v[7]);
        lds[(4 * i + 2) * STRIDE] = make_uint4(f.v[8], f.v[9], f.v[10], f.v[11]);
        lds[(4 * i + 3) * STRIDE] = make_uint4(f.v[12], f.v[13], f.v[14], f.v[15]);
    }
    __device__ __forceinline__ fe get_piece(int i) const {
        return fe_from_u4(lds[(4 * i + 0) * STRIDE], lds[(4 * i + 1) * STRIDE], lds[(4 * i + 2) * STRIDE], lds[(4 * i + 3) * STRIDE]);
    }
    __device__ __forceinline__ void put_step(const pniels &e) const {
        put_piece(0, e.a);
        put_piece(1, e.b);
        put_piece(2, e.cn);
        put_piece(3, e.z);
    }
    __device__ __forceinline__ pniels step() const {
        // read anew in every iteration of the build: hoisted out of the loop (nothing there writes LDS) the step
        // would sit in 64 registers next to the accumulator, which is exactly what spills
        asm volatile("" ::: "memory");
        pniels e;
        e.a = get_piece(0);
        e.b = get_piece(1);
        e.cn = get_piece(2);
        e.z = get_piece(3);
        return e;
    }
};
// eddsa.hpp's decode_into_table for these tables, out of line: k_ed448_verify calls it for the key and for R instead of
// carrying two inlined copies (62 K -> 49 K instructions), and the register allocation of the decoding's
// exponentiation no longer shares a function with the hash and the ladder: the kernel's .vgpr_spill_count goes
// 422 -> 44; the function itself moves 148 registers to scratch and back, once per call, the caller's live state
// among them.  Same time with pooled keys, 1 % less when every lane decodes its own key (profiles/r03/experiments.md B3).
__device__ __noinline__ static bool decode_into_table(LdsStepTable<> tab, const uint8_t *enc, bool negate) {
    return decode_into_table<LdsStepTable<>>(tab, enc, negate);
}
// The comb staged in LDS and gathered with wavefront shuffles.  Entry e occupies words
// [49e, 49e+48) (stride 49 keeps the fill reads below conflict-free).  For comb j every lane
// first reads 12 words with a LANE-dependent, index-INDEPENDENT address: lane l takes words
// 4q + (l>>4), q < 12, of entry 16j + (l&15), so the wave's registers hold the whole 16-entry
// sub-table once.  Word 4q+c of the entry a lane wants then comes from lane idx + 16c through
// ds_bpermute_b32: the crossbar has no bank conflicts, so neither the addresses issued nor the
// time taken depend on the (possibly secret) digit.
constexpr int COMB_LDS_STRIDE = 49;
constexpr int COMB_LDS_WORDS = 80 * COMB_LDS_STRIDE;
struct LdsShuffleComb {
    using plan = comb_ref;
    const uint32_t *lds;
    uint32_t lane;  // lane within the wave
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint32_t *src = lds + (16 * j + (lane & 15)) * COMB_LDS_STRIDE + (lane >> 4);
        uint32_t r[12];
#pragma unroll
        for (int q = 0; q < 12; q++) r[q] = src[4 * q];
        const int a0 = (int)(idx << 2);  // bpermute takes a byte address: 4 * source lane
        niels e;
        uint32_t w[48];
#pragma unroll
        for (int q = 0; q < 12; q++) {
#pragma unroll
            for (int c = 0; c < 4; c++) w[4 * q + c] = (uint32_t)__builtin_amdgcn_ds_bpermute(a0 + 64 * c, (int)r[q]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            e.a.v[i] = w[i];
            e.b.v[i] = w[16 + i];
            e.cn.v[i] = w[32 + i];
        }
        return e;
    }
};
template <int ENTRIES = 80>
__device__ __forceinline__ void stage_comb_lds(uint32_t *lds, const uint4 *comb) {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(comb);
    for (int i = threadIdx.x; i < ENTRIES * 48; i += BLOCK) lds[(i / 48) * COMB_LDS_STRIDE + (i % 48)] = g[i];
    __syncthreads();
}
// The same for the library's own 4 x 7 x 16 comb of the base point (scalarmul.hpp comb_big; built once per
// device by k_build_comb_big): a comb has 64 entries, so lane l reads ALL 48 words of entry 64j + l -- a
// lane-dependent, index-independent address, conflict-free with the odd stride -- and the wanted entry's
// words come from lane idx through ds_bpermute_b32.
constexpr int COMB_BIG_LDS_WORDS = comb_big::ENTRIES * COMB_LDS_STRIDE;
struct LdsShuffleCombBig {
    using plan = comb_big;
    const uint32_t *lds;
    uint32_t lane;  // lane within the wave
    __device__ __forceinline__ fe gather(const uint32_t *src, int a0) const {
        uint32_t r[16];
#pragma unroll
        for (int q = 0; q < 16; q++) r[q] = src[q];
        fe v;
#pragma unroll
        for (int q = 0; q < 16; q++) v.v[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(a0, (int)r[q]);
        return v;
    }
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint32_t *src = lds + (64 * j + lane) * COMB_LDS_STRIDE;
        const int a0 = (int)(idx << 2);  // bpermute takes a byte address: 4 * source lane
        niels e;
        e.a = gather(src, a0);
        e.b = gather(src + 16, a0);
        e.cn = gather(src + 32, a0);
        return e;
    }
};

// Fixed-base window table of the base point: a header of BWT_HEADER_U4 uint4 -- [0] = (digit bits, windows, 0, 0),
// [1..4] = the recoding offset (2^(bits*windows) - 1) mod q -- and windows x 2^(bits-1) affine niels (12 uint4 each) in
// global memory, built once per device at its first use (k_bwt_header, k_bwt_steps, k_build_bwt).  16-bit digits keep
// it in the Infinity Cache (168 MiB); wider ones trade HBM for additions (24 bits: 28.5 GiB, 18 additions instead of
// 27): every lane gathers one contiguous 192-byte entry per digit, requested one addition ahead, and the kernels that
// use it do not wait for it (profiles/r04/experiments.md I).
constexpr int BWT_HEADER_U4 = 16;
struct GlobalBwt {
    const uint4 *p;
    __device__ __forceinline__ BwtGeom geom() const {
        const uint4 h = p[0];
        return BwtGeom{h.x, h.y};
    }
    __device__ __forceinline__ sc adjust() const {
        const uint4 a = p[1], b = p[2], c = p[3], d = p[4];
        sc r;
        r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
        r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
        r.w[8] = c.x; r.w[9] = c.y; r.w[10] = c.z; r.w[11] = c.w;
        r.w[12] = d.x; r.w[13] = d.y;
        return r;
    }
    __device__ __forceinline__ niels load(const BwtGeom &g, uint32_t i, uint32_t idx) const {
        const uint4 *q = p + BWT_HEADER_U4 + 12 * (((size_t)i << (g.bits - 1)) + idx);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};
// A 4 x 7 x 16 comb in global memory, read by the digit (public scalars only): the comb of a verification key that
// signed many of a batch's signatures (kernels_verify.hip), 256 affine niels = 48 KiB per key.
constexpr int KEY_TEETH_U4 = 2 * comb_xwide::TEETH * comb_xwide::COMBS * 16;   // room for 45 teeth + their doubles (pniels) per key while its comb is built
constexpr int KEY_COMB_U4 = comb_big::ENTRIES * 12;
// a key's comb has 7, 8 or 9 teeth per comb (ctrl[3], k_verify_key_mode): 4 x 7 x 16, 4 x 8 x 14 or 5 x 9 x 10 -- its
// combs, their spacing, entries and uint4
__host__ __device__ constexpr uint32_t key_comb_combs(uint32_t teeth) { return teeth >= (uint32_t)comb_xwide::TEETH ? (uint32_t)comb_xwide::COMBS : 4u; }
__host__ __device__ constexpr uint32_t key_comb_spacing(uint32_t teeth) {
    return teeth >= (uint32_t)comb_xwide::TEETH ? (uint32_t)comb_xwide::SPACING : 448u / (4u * teeth);
}
__host__ __device__ constexpr uint32_t key_comb_entries(uint32_t teeth) { return key_comb_combs(teeth) << (teeth - 1); }
__host__ __device__ constexpr uint32_t key_comb_u4(uint32_t teeth) { return 12u * key_comb_entries(teeth); }

constexpr int KEY_COMBS_MAX = 1 << 17;     // the most keys of a batch that can have combs (62 KiB of workspace each)
constexpr int KEY_COMBS_MIN_BATCH = 4096;   // combs are considered from so many signatures on (up to there a wave verifies each signature, section 7a)
constexpr int KEY_SORT_BINS = 8192;        // up to so many keys the counting sort goes through per-block bins in LDS
constexpr int KEY_TEETH_BY_WAVE_MAX = 4096;   // up to so many keys a WAVE computes a key's teeth (latency), beyond a lane (throughput)
// the fewest entries of a key's comb that one lane of k_verify_key_combs builds and normalises with one shared inversion,
// the most, and the share of the resident lanes that its segments may be (kernels_verify.hip key_comb_segment picks
// the length on the device from the number of keys; profiles/r04/experiments.md H, profiles/r06/combsphases.txt)
#ifndef GD_KEY_COMB_SEG
#define GD_KEY_COMB_SEG 8
#endif
#ifndef GD_KEY_COMB_SEG_MAX
#define GD_KEY_COMB_SEG_MAX 64
#endif
constexpr int KEY_COMB_SEG = GD_KEY_COMB_SEG, KEY_COMB_SEG_MAX = GD_KEY_COMB_SEG_MAX;   // (a comb of 7 teeth has 64 entries)
#ifndef GD_KEY_COMB_OCC_NUM
#define GD_KEY_COMB_OCC_NUM 1
#endif
#ifndef GD_KEY_COMB_OCC_DEN
#define GD_KEY_COMB_OCC_DEN 2
#endif
constexpr uint32_t KEY_COMB_OCC_NUM = GD_KEY_COMB_OCC_NUM, KEY_COMB_OCC_DEN = GD_KEY_COMB_OCC_DEN;
static_assert(KEY_COMB_SEG >= 1 && KEY_COMB_SEG <= KEY_COMB_SEG_MAX && KEY_COMB_SEG_MAX <= 64, "a comb of 7 teeth has 64 entries");
constexpr int KEYCOMB_SLOT_U4 = 16;   // what a verification parks until its lane's shared inversion (kernels_verify.hip)
// the launches of one group of key-comb verifications (one shared inversion per lane over all of them): where each
// launch's positions start, relative to the group's, and how many they are
constexpr int VERIFY_CHUNKS_MAX = 16;
struct VerifyChunks {
    uint32_t count;
    uint32_t lo[VERIFY_CHUNKS_MAX], m[VERIFY_CHUNKS_MAX];
};
template <class PLAN>
struct GlobalCombOf {
    using plan = PLAN;
    const uint4 *p;
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint4 *q = p + 12 * (PLAN::PER_COMB * j + idx);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};
using GlobalCombBig = GlobalCombOf<comb_big>;

struct LdsStage {  // 136-byte sponge block per lane, word-interleaved across lanes
    uint32_t *p;   // &stage[threadIdx.x]
    __device__ __forceinline__ void put(uint32_t i, uint32_t b) const {
        reinterpret_cast<uint8_t *>(p + (i >> 2) * BLOCK)[i & 3] = (uint8_t)b;
    }
    __device__ __forceinline__ uint64_t get64(int k) const {
        return (uint64_t)p[(2 * k) * BLOCK] | (uint64_t)p[(2 * k + 1) * BLOCK] << 32;
    }
};
struct LdsMkBits {
    uint32_t *slot0;  // two scalar slots, 15*BLOCK words apart
    __device__ __forceinline__ LdsBits operator()(const sc &s, int which) const {
        return lds_put_bits(slot0 + which * 15 * BLOCK, s);
    }
    __device__ __forceinline__ LdsBits words(const uint32_t (&w)[15], int which) const {
        uint32_t *slot = slot0 + which * 15 * BLOCK;
#pragma unroll
        for (int k = 0; k < 15; k++) slot[k * BLOCK] = w[k];
        return LdsBits{slot};
    }
};

// Verification's three scalars in 16 words of LDS per lane: the two half-size ones of the joint ladder are 225
// bits each (words 0..7 and 8..15), and the base point's full-size one is only made when they are dead (words 0..13
// and a zero fifteenth).
constexpr int VERIFY_LDS_WORDS = 32;   // per lane: 16 for the scalars' bits, 32 where a key's comb is walked by its digits
struct LdsMkBitsVerify {
    uint32_t *slot0;
    __device__ __forceinline__ LdsBits operator()(const sc &s, int) const {
#pragma unroll
        for (int k = 0; k < 14; k++) slot0[k * BLOCK] = s.w[k];
        slot0[14 * BLOCK] = 0;   // (window_bwt reads a digit's second word whether it straddles or not)
        return LdsBits{slot0};
    }
    __device__ __forceinline__ LdsBits words(const uint32_t (&w)[15], int which) const {
        uint32_t *slot = slot0 + which * 8 * BLOCK;
#pragma unroll
        for (int k = 0; k < 8; k++) slot[k * BLOCK] = w[k];
        return LdsBits{slot};
    }
    // the digits of a key's comb in the order its walk consumes them (scalarmul.hpp comb_digits): two per word, the
    // same word-major slot -- VERIFY_LDS_WORDS per lane cover the widest plan's 64 digits
    // (get() re-derives the lane's slot from the wave's first thread -- a scalar register -- and the lane number: kept in
    // a vector register across the walk's loop, the slot's address was spilled and re-read from scratch, waited for,
    // ahead of every digit)
    struct Digits {
        uint32_t *slot;
        uint32_t wave_slot;     // LDS byte address of the slot of the wave's lane 0: uniform
        __device__ __forceinline__ void put(int k, uint32_t two) const { slot[k * BLOCK] = two; }
        __device__ __forceinline__ uint32_t get(int t) const {
            uint32_t lane;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
            const uint32_t at = wave_slot + 4u * lane + (uint32_t)(t >> 1) * (4u * BLOCK) + 2u * (uint32_t)(t & 1);
            return *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)at;
        }
    };
    template <class PLAN>
    __device__ __forceinline__ Digits digits(const sc &recoded) const {
        static_assert(comb_digits<PLAN>::WORDS <= VERIFY_LDS_WORDS, "the lane's LDS slot holds the plan's digits");
        const uint32_t in_wave = threadIdx.x & 63u;
        Digits d{slot0, (uint32_t)__builtin_amdgcn_readfirstlane(
                            (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(slot0 - in_wave))};
        comb_digits<PLAN>::store(d, recoded);
        return d;
    }
};

#define GD_KERNEL extern "C" __global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)

// ---------------------------------------------------------------- kernel prototypes
// Definitions live in kernels_{varbase,verify,fixed,misc}.hip (separate translation units so the
// build compiles them in parallel); the host runtime (goldilocks_amd.hip) launches them.
// (no __restrict__ on an output the host runtime aliases with an input: out/base, out/b2, out1/base, out/a)
GD_KERNEL k_point_scalarmul_ct(uint64_t *out, const uint64_t *base, const uint64_t *__restrict__ scalar,
                                  uint32_t n, uint4 *__restrict__ workspace);
// one operation per WAVE (wave_coop.hpp): the small-batch / single-call path
extern "C" __global__ void k_point_scalarmul_wave(uint64_t *out, const uint64_t *base, const uint64_t *__restrict__ scalar,
                                                  uint32_t n);
extern "C" __global__ void k_double_scalarmul_wave(uint64_t *out, const uint64_t *b1, const uint64_t *__restrict__ s1,
                                                   const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                                                   const uint4 *__restrict__ bwt);
extern "C" __global__ void k_x448_wave(uint8_t *__restrict__ shared, int32_t *__restrict__ status, const uint8_t *__restrict__ base,
                                       const uint8_t *__restrict__ scalar, uint32_t n);
extern "C" __global__ void k_precomputed_scalarmul_wave(uint64_t *__restrict__ out, const uint4 *__restrict__ comb,
                                                        const uint64_t *__restrict__ scalar, uint32_t n);
extern "C" __global__ void k_derive_wave(uint8_t *__restrict__ out, const uint8_t *__restrict__ sk, uint32_t n,
                                         const uint4 *__restrict__ comb, int x448_keygen);
extern "C" __global__ void k_ed448_sign_wave(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                                             const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                                             uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                                             uint32_t n, const uint4 *__restrict__ comb);
extern "C" __global__ void k_point_encode_wave(uint8_t *__restrict__ ser, const uint64_t *__restrict__ pts, uint32_t n, int eddsa);
extern "C" __global__ void k_point_decode_wave(uint64_t *__restrict__ pts, int32_t *__restrict__ status,
                                               const uint8_t *__restrict__ ser, uint32_t n, int eddsa, int allow_identity);
extern "C" __global__ void k_point_from_hash_wave(uint64_t *__restrict__ out, const uint8_t *__restrict__ hash, uint32_t n,
                                                  int uniform);
extern "C" __global__ void k_precompute_wave(uint64_t *__restrict__ tables, const uint64_t *__restrict__ base, uint32_t n);
extern "C" __global__ void k_direct_scalarmul_wave(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                                                   const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar,
                                                   uint32_t n, int allow_identity, int short_circuit,
                                                   const uint64_t *__restrict__ point_base_abi);
extern "C" __global__ void k_point_dual_scalarmul_wave(uint64_t *a1, uint64_t *a2, const uint64_t *base,
                                                       const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2,
                                                       uint32_t n);
extern "C" __global__ void k_wave_field_op(uint64_t *__restrict__ out, int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                                           const uint64_t *__restrict__ b, uint32_t n, int op);
extern "C" __global__ void k_ed448_verify_wave(int32_t *__restrict__ status, const uint8_t *__restrict__ sig,
                                               const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                                               const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                               const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                                               const uint4 *__restrict__ bwt);
GD_KERNEL k_precomputed_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ comb,
                                  const uint64_t *__restrict__ scalar, uint32_t n);
GD_KERNEL k_base_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ bwt,
                           const uint64_t *__restrict__ scalar, uint32_t n);
GD_KERNEL k_double_scalarmul(uint64_t *out, const uint64_t *b1, const uint64_t *__restrict__ s1,
                             const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                             uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt);
GD_KERNEL k_double_scalarmul_ct(uint64_t *out, const uint64_t *b1, const uint64_t *__restrict__ s1,
                                   const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                                   uint4 *__restrict__ workspace, const uint64_t *__restrict__ point_base_abi);
GD_KERNEL k_point_encode_eddsa_shared(uint8_t *__restrict__ enc, const uint64_t *__restrict__ pts, uint32_t n,
                                      uint4 *__restrict__ ws);
GD_KERNEL k_build_comb_big(uint4 *__restrict__ dst, const uint4 *__restrict__ comb);
GD_KERNEL k_base_scalarmul_ct(uint64_t *__restrict__ out, const uint4 *__restrict__ comb_big_tab,
                              const uint64_t *__restrict__ scalar, uint32_t n, uint32_t halve);
GD_KERNEL k_recomb_big(uint4 *__restrict__ dst, const uint4 *__restrict__ comb);
GD_KERNEL k_half_size_pair(uint32_t *__restrict__ rho, uint32_t *__restrict__ tau, const uint64_t *__restrict__ h,
                           uint32_t n);
GD_KERNEL k_ed448_verify(int32_t *__restrict__ status, const uint8_t *__restrict__ sig,
                         const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                         const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                         const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                         uint4 *__restrict__ workspace, const uint4 *__restrict__ bwt,
                         const uint32_t *__restrict__ rep, const uint32_t *__restrict__ slot_of,
                         const uint4 *__restrict__ pool, const uint8_t *__restrict__ key_ok,
                         const uint32_t *__restrict__ ctrl);
// one decoding and one window table per distinct public key of a verification batch (kernels_verify.hip)
GD_KERNEL k_verify_dedupe(uint32_t *__restrict__ rep, uint32_t *__restrict__ slot_of, uint32_t *__restrict__ key_list,
                          uint32_t *__restrict__ hash_slots, uint32_t hash_mask, uint32_t *__restrict__ ctrl,
                          const uint8_t *__restrict__ pk, uint32_t n, uint32_t seed);
GD_KERNEL k_verify_key_mode(uint32_t *__restrict__ ctrl, uint32_t n, uint32_t pool_capacity, uint32_t comb_capacity,
                            uint32_t comb_min_per_key, uint32_t wide_min_per_key, uint32_t xwide_min_per_key);
GD_KERNEL k_verify_key_tables(uint4 *__restrict__ pool, uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                              const uint32_t *__restrict__ key_list, const uint8_t *__restrict__ pk);
extern "C" __global__ void k_verify_key_teeth(uint4 *__restrict__ teeth, uint8_t *__restrict__ key_ok,
                                              const uint32_t *__restrict__ ctrl, const uint32_t *__restrict__ key_list,
                                              const uint8_t *__restrict__ pk);
GD_KERNEL k_verify_key_teeth_lanes(uint4 *__restrict__ teeth, uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                   const uint32_t *__restrict__ key_list, const uint8_t *__restrict__ pk);
GD_KERNEL k_verify_key_combs(uint4 *__restrict__ combs, const uint4 *__restrict__ teeth, const uint32_t *__restrict__ ctrl,
                             uint4 *__restrict__ chain);
GD_KERNEL k_ed448_verify_keycomb(const uint8_t *__restrict__ sig,
                                 const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                                 const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                 const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                                 const uint4 *__restrict__ bwt, const uint32_t *__restrict__ rep,
                                 const uint32_t *__restrict__ slot_of, const uint4 *__restrict__ combs,
                                 const uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                 uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                 uint4 *__restrict__ chain_state, uint32_t resume,
                                 const uint4 *__restrict__ qpark, uint32_t q_count, int32_t *__restrict__ finish_status);
GD_KERNEL k_ed448_verify_keycomb_wide(const uint8_t *__restrict__ sig,
                                 const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                                 const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                 const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                                 const uint4 *__restrict__ bwt, const uint32_t *__restrict__ rep,
                                 const uint32_t *__restrict__ slot_of, const uint4 *__restrict__ combs,
                                 const uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                 uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                 uint4 *__restrict__ chain_state, uint32_t resume,
                                 const uint4 *__restrict__ qpark, uint32_t q_count, int32_t *__restrict__ finish_status);
GD_KERNEL k_ed448_verify_keycomb_xwide(const uint8_t *__restrict__ sig,
                                 const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
                                 const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
                                 const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n,
                                 const uint4 *__restrict__ bwt, const uint32_t *__restrict__ rep,
                                 const uint32_t *__restrict__ slot_of, const uint4 *__restrict__ combs,
                                 const uint8_t *__restrict__ key_ok, const uint32_t *__restrict__ ctrl,
                                 uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                 uint4 *__restrict__ chain_state, uint32_t resume,
                                 const uint4 *__restrict__ qpark, uint32_t q_count, int32_t *__restrict__ finish_status);
GD_KERNEL k_verify_base_part(uint4 *__restrict__ qpark, const uint8_t *__restrict__ sig, const uint32_t *__restrict__ order,
                             uint32_t q_count, const uint4 *__restrict__ bwt, const uint32_t *__restrict__ ctrl);
GD_KERNEL k_ed448_verify_keycomb_finish(int32_t *__restrict__ status, const uint32_t *__restrict__ ctrl,
                                        const uint4 *__restrict__ park, const uint32_t *__restrict__ order,
                                        const uint4 *__restrict__ chain_state, VerifyChunks chunks);
GD_KERNEL k_verify_key_count(uint32_t *__restrict__ count, const uint32_t *__restrict__ rep,
                             const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ ctrl, uint32_t n);
GD_KERNEL k_verify_key_scan(uint32_t *__restrict__ count, const uint32_t *__restrict__ ctrl);
GD_KERNEL k_verify_key_scatter(uint32_t *__restrict__ order, uint32_t *__restrict__ count, const uint32_t *__restrict__ rep,
                               const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ ctrl, uint32_t n,
                               uint32_t base);
GD_KERNEL k_ed448_derive_public_key(uint8_t *__restrict__ pk, const uint8_t *__restrict__ sk, uint32_t n,
                                    const uint4 *__restrict__ bwt, uint4 *__restrict__ workspace);
GD_KERNEL k_ed448_sign(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                       const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                       uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                       uint32_t n, const uint4 *__restrict__ bwt, uint4 *__restrict__ workspace);
// index-independent variants (comb in LDS + wavefront-shuffle gather) of the three kernels that
// multiply the base point by a SECRET scalar; selected by goldilocks_amd_set_table_access()
GD_KERNEL k_ed448_derive_public_key_ct(uint8_t *__restrict__ pk, const uint8_t *__restrict__ sk, uint32_t n,
                                       const uint4 *__restrict__ comb, uint4 *__restrict__ workspace);
GD_KERNEL k_ed448_sign_ct(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                          const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                          uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                          uint32_t n, const uint4 *__restrict__ comb, uint4 *__restrict__ workspace);
GD_KERNEL k_x448_derive_ct(uint8_t *__restrict__ shared, const uint8_t *__restrict__ scalar, uint32_t n,
                           const uint4 *__restrict__ comb, uint4 *__restrict__ workspace);
GD_KERNEL k_direct_scalarmul_ct(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                                const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar, uint32_t n,
                                int allow_identity, int short_circuit, const uint64_t *__restrict__ point_base_abi);
GD_KERNEL k_point_dual_scalarmul_ct(uint64_t *out1, uint64_t *out2, const uint64_t *base,
                                       const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2, uint32_t n,
                                       uint4 *__restrict__ workspace);
GD_KERNEL k_point_from_hash(uint64_t *__restrict__ out, const uint8_t *__restrict__ hash, uint32_t n, int uniform);
GD_KERNEL k_x448(uint8_t *__restrict__ shared, int32_t *__restrict__ status, const uint8_t *__restrict__ base,
                 const uint8_t *__restrict__ scalar, uint32_t n, const uint4 *__restrict__ bwt,
                 uint4 *__restrict__ workspace);
GD_KERNEL k_point_encode(uint8_t *__restrict__ ser, const uint64_t *__restrict__ pts, uint32_t n, int eddsa);
GD_KERNEL k_bwt_export(uint8_t *__restrict__ out, const uint4 *__restrict__ bwt, uint64_t first, uint32_t count);
GD_KERNEL k_ed448_expand_secret(uint8_t *__restrict__ out, const uint8_t *__restrict__ sk, uint32_t n, int as_scalar);
GD_KERNEL k_scalar_op(uint64_t *out, int32_t *__restrict__ status, const uint8_t *a, const uint64_t *b, uint32_t n, int op, uint32_t len);
GD_KERNEL k_x448_from_edwards(uint8_t *__restrict__ out, const uint8_t *__restrict__ ed, const uint64_t *__restrict__ pts,
                              uint32_t n);
GD_KERNEL k_point_decode(uint64_t *__restrict__ pts, int32_t *__restrict__ status,
                         const uint8_t *__restrict__ ser, uint32_t n, int eddsa, int allow_identity);
GD_KERNEL k_point_op(uint64_t *out, const uint64_t *a, const uint64_t *__restrict__ b,
                     uint32_t n, int op);
GD_KERNEL k_point_pred(int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                       const uint64_t *__restrict__ b, uint32_t n, int op);
GD_KERNEL k_field_op(uint64_t *__restrict__ out, int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                     const uint64_t *__restrict__ b, uint32_t n, int op, uint32_t aux);
GD_KERNEL k_import_comb(uint4 *__restrict__ dst, const uint64_t *__restrict__ src, uint32_t ntables);
// the base point's window table (kernels_fixed.hip): header, the windows' steps 2 * 2^(bits i) * B, the entries in
// segments of BWT_BUILD_SEG per lane (slab: the entries [first, first + count) of the whole table; chain: a field
// element of scratch per entry of the slab)
constexpr uint32_t BWT_BUILD_SEG = 64;
GD_KERNEL k_bwt_header(uint4 *__restrict__ table, uint32_t bits);
GD_KERNEL k_bwt_steps(uint4 *__restrict__ steps, const uint4 *__restrict__ comb, uint32_t bits);
GD_KERNEL k_build_bwt(uint4 *__restrict__ table, const uint4 *__restrict__ comb, const uint4 *__restrict__ steps,
                      uint4 *__restrict__ chain, uint32_t bits, uint64_t first, uint32_t count);
GD_KERNEL k_precompute(uint64_t *__restrict__ tables, const uint64_t *__restrict__ base, uint32_t n,
                       uint4 *__restrict__ workspace);

}  // namespace gd
