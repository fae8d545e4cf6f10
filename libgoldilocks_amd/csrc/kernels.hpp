// kernels.hpp -- HIP kernels for gfx950: one Ed448 operation per wavefront lane.
//
// Memory plan:
//   * I/O arrays are the reference's AoS structs (256-B points, 56-B scalars); each lane
//     moves its own struct with 16-byte vector accesses, every byte of every fetched
//     line is used.  I/O is ~568 B per ~4000 field multiplications: not the bound.
//   * recoded scalars live in LDS, word-major ([word][lane]): window/comb bit positions
//     are wave-uniform, so every LDS read is conflict-free.
//   * the per-lane window table of a variable base (16 projective niels = 4 KiB) cannot
//     fit LDS (64 lanes x 4 KiB = 256 KiB per wave), so it lives in an HBM workspace,
//     lane-contiguous, sized by RESIDENT lanes (grid-stride), read 256 B per window.
//   * shared read-only tables (base-point comb / window table) sit in one small global
//     buffer that stays L1/L2 resident.
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
constexpr int PRECOMP_U4 = 80 * 20 + 64;  // uint4 per lane for k_precompute: 80 x 5 fe + 4 doubled teeth

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

struct LaneTable {  // this lane's 16-entry window table in the HBM workspace
    uint4 *p;
    __device__ __forceinline__ void store(int k, const pniels &e) const {
        uint4 *q = p + 16 * k;
        fe_store(q, e.a);
        fe_store(q + 4, e.b);
        fe_store(q + 8, e.cn);
        fe_store(q + 12, e.z);
    }
    __device__ __forceinline__ pniels load(uint32_t k) const {
        const uint4 *q = p + 16 * k;
        pniels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        e.z = fe_load(q + 12);
        return e;
    }
};
struct SharedTable {  // read-only 16-entry table shared by all lanes (base point)
    const uint4 *p;
    __device__ __forceinline__ pniels load(uint32_t k) const {
        const uint4 *q = p + 16 * k;
        pniels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        e.z = fe_load(q + 12);
        return e;
    }
};
struct SharedComb {  // 80 affine niels, 12 uint4 each, our limb/sign convention
    const uint4 *p;
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint4 *q = p + 12 * (16 * j + idx);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};
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
__device__ __forceinline__ void stage_comb_lds(uint32_t *lds, const uint4 *comb) {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(comb);
    for (int i = threadIdx.x; i < 80 * 48; i += BLOCK) lds[(i / 48) * COMB_LDS_STRIDE + (i % 48)] = g[i];
    __syncthreads();
}

// Fixed-base window table of the base point: 56 x 128 affine niels (12 uint4 each), 1.3 MiB in
// global memory, built once per device (k_build_bwt) and L2-resident: every lane gathers one
// contiguous 192-byte entry per digit.
constexpr int BWT_ENTRIES = 56 * 128;
struct GlobalBwt {
    const uint4 *p;
    __device__ __forceinline__ niels load(int i, uint32_t idx) const {
        const uint4 *q = p + 12 * (128 * i + idx);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};

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
};

#define GD_KERNEL extern "C" __global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)

// ---------------------------------------------------------------- kernels

// config 2: scaled[i] = scalar[i] * base[i]   (ref: goldilocks_448_point_scalarmul)
GD_KERNEL k_point_scalarmul(uint64_t *__restrict__ out, const uint64_t *__restrict__ base,
                            const uint64_t *__restrict__ scalar, uint32_t n, uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        pt b = pt_load_abi(base + 32 * (size_t)i);
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(scalar + 7 * (size_t)i)));
        build_window_table(tab, b);
        pt r = ladder_varbase(bits, tab);
        pt_store_abi(out + 32 * (size_t)i, r);
    }
}

// config 3: scaled[i] = scalar[i] * G, G given by a comb table   (ref: goldilocks_448_precomputed_scalarmul)
GD_KERNEL k_precomputed_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ comb,
                                  const uint64_t *__restrict__ scalar, uint32_t n) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    // every lane of a wave must take part in the shuffles: iterate wave-uniformly and clamp
    const uint32_t rounds = (n + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t i_raw = lane + r * stride;
        const bool live = i_raw < n;
        const uint32_t i = live ? i_raw : n - 1;
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(scalar + 7 * (size_t)i)));
        pt res = ladder_comb(bits, tab);
        if (live) pt_store_abi(out + 32 * (size_t)i, res);
    }
}

// scaled[i] = scalar[i] * B for the built-in base point, through the 8-bit window table
GD_KERNEL k_base_scalarmul(uint64_t *__restrict__ out, const uint4 *__restrict__ bwt,
                           const uint64_t *__restrict__ scalar, uint32_t n) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt tab{bwt};
    for (uint32_t i = lane; i < n; i += stride) {
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed8(sc_load_abi(scalar + 7 * (size_t)i)));
        pt_store_abi(out + 32 * (size_t)i, ladder_bwt(bits, tab));
    }
}

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i]; b1 == nullptr: b1 is the base point (shared table)
GD_KERNEL k_double_scalarmul(uint64_t *__restrict__ out, const uint64_t *__restrict__ b1,
                             const uint64_t *__restrict__ s1, const uint64_t *__restrict__ b2,
                             const uint64_t *__restrict__ s2, uint32_t n, uint4 *__restrict__ workspace,
                             const uint4 *__restrict__ base_tab) {
    __shared__ uint32_t s_bits[30 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable t2{workspace + (size_t)lane * 2 * TABLE_U4};
    LaneTable t1{workspace + (size_t)lane * 2 * TABLE_U4 + TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        LdsBits bits1 = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(s1 + 7 * (size_t)i)));
        LdsBits bits2 =
            lds_put_bits(s_bits + 15 * BLOCK + threadIdx.x, sc_recode_signed(sc_load_abi(s2 + 7 * (size_t)i)));
        build_window_table(t2, pt_load_abi(b2 + 32 * (size_t)i));
        pt r;
        if (b1) {  // uniform
            build_window_table(t1, pt_load_abi(b1 + 32 * (size_t)i));
            r = ladder_double(bits1, t1, bits2, t2);
        } else {
            r = ladder_double(bits1, SharedTable{base_tab}, bits2, t2);
        }
        pt_store_abi(out + 32 * (size_t)i, r);
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
    LaneTable a_tab{workspace + (size_t)lane * TABLE_U4};
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> b_tab{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    for (uint32_t i = lane; i < n; i += stride) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint32_t mlen = msg_offsets ? (uint32_t)(msg_offsets[i + 1] - msg_offsets[i]) : msg_len;
        Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx,
                                            ctx_len);
        bool ok = ed448_verify_core(m, b_tab, a_tab, stage, mk);
        status[i] = ok ? -1 : 0;
    }
}

// "next" row f1: pk[i] = derive_public_key(sk[i])   (ref: goldilocks_ed448_derive_public_key)
GD_KERNEL k_ed448_derive_public_key(uint8_t *__restrict__ pk, const uint8_t *__restrict__ sk, uint32_t n,
                                    const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> fb{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    for (uint32_t i = lane; i < n; i += stride)
        ed448_derive_core(pk + 57 * (size_t)i, sk + 57 * (size_t)i, fb, stage, mk);
}

// "next" row f1: sig[i] = sign(sk[i], pk[i], msg[i])   (ref: goldilocks_ed448_sign)
GD_KERNEL k_ed448_sign(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                       const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                       uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                       uint32_t n, const uint4 *__restrict__ bwt, uint8_t *__restrict__ workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> fb{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBits mk{s_bits + threadIdx.x};
    uint8_t *scratch = workspace + (size_t)lane * 64;   // the hashed-key seed of the signature in flight
    for (uint32_t i = lane; i < n; i += stride) {
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint32_t mlen = msg_offsets ? (uint32_t)(msg_offsets[i + 1] - msg_offsets[i]) : msg_len;
        ed448_sign_core(sig + 114 * (size_t)i, sk + 57 * (size_t)i, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx,
                        ctx_len, scratch, fb, stage, mk);
    }
}

// "next" row f2: wire-format scalarmul, 56 bytes in / 56 bytes out   (ref: goldilocks_448_direct_scalarmul)
GD_KERNEL k_direct_scalarmul(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                             const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar, uint32_t n,
                             int allow_identity, int short_circuit, uint4 *__restrict__ workspace,
                             const uint64_t *__restrict__ point_base_abi) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        uint32_t w[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        pt b;
        bool ok = pt_decode_words(b, w, allow_identity != 0);
        status[i] = ok ? -1 : 0;
        if (!ok && short_circuit) continue;
        if (!ok) b = pt_load_abi(point_base_abi);   // src/goldilocks.c:898: multiply the base point instead
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(scalar + 7 * (size_t)i)));
        build_window_table(tab, b);
        pt r = ladder_varbase(bits, tab);
        pt_encode_words(w, r);
        uint32_t *dst = reinterpret_cast<uint32_t *>(scaled + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = w[k];
    }
}

// "next" row f4: (s1*B, s2*B) for one base   (ref: goldilocks_448_point_dual_scalarmul)
GD_KERNEL k_point_dual_scalarmul(uint64_t *__restrict__ out1, uint64_t *__restrict__ out2,
                                 const uint64_t *__restrict__ base, const uint64_t *__restrict__ s1,
                                 const uint64_t *__restrict__ s2, uint32_t n, uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[30 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        LdsBits b1 = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(s1 + 7 * (size_t)i)));
        LdsBits b2 = lds_put_bits(s_bits + 15 * BLOCK + threadIdx.x, sc_recode_signed(sc_load_abi(s2 + 7 * (size_t)i)));
        build_window_table(tab, pt_load_abi(base + 32 * (size_t)i));
        pt r1, r2;
        ladder_dual(r1, r2, b1, b2, tab);
        pt_store_abi(out1 + 32 * (size_t)i, r1);
        pt_store_abi(out2 + 32 * (size_t)i, r2);
    }
}

// "next" row f4: Elligator 2 hash-to-curve   (ref: goldilocks_448_point_from_hash_nonuniform / _uniform)
GD_KERNEL k_point_from_hash(uint64_t *__restrict__ out, const uint8_t *__restrict__ hash, uint32_t n, int uniform) {
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t nb = uniform ? 112 : 56;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint32_t w[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(hash + (size_t)nb * i);   // 56 | nb: 8-byte aligned
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        pt p = pt_from_hash_words(w);
        if (uniform) {
#pragma unroll
            for (int k = 0; k < 14; k++) w[k] = src[14 + k];
            p = pt_add(p, pt_from_hash_words(w), false);
        }
        pt_store_abi(out + 32 * (size_t)i, p);
    }
}

// "next" row f3: X448.  base == nullptr: derive_public_key through the comb   (ref: goldilocks_x448*)
GD_KERNEL k_x448(uint8_t *__restrict__ shared, int32_t *__restrict__ status, const uint8_t *__restrict__ base,
                 const uint8_t *__restrict__ scalar, uint32_t n, const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt tab{bwt};
    for (uint32_t i = lane; i < n; i += stride) {
        const bool live = true;
        uint32_t w[14], o[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(scalar + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        bool ok = true;
        if (base) {
            uint32_t b[14];
            const uint32_t *bs = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
            for (int k = 0; k < 14; k++) b[k] = bs[k];
            sc raw;
#pragma unroll
            for (int k = 0; k < 14; k++) raw.w[k] = w[k];
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, raw);
            ok = x448_core(o, b, bits);
        } else {
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed8(x448_public_scalar(w)));
            pt_encode_x448_words(o, ladder_bwt(bits, tab));
        }
        if (live) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(shared + 56 * (size_t)i);
#pragma unroll
            for (int k = 0; k < 14; k++) dst[k] = o[k];
            if (status) status[i] = ok ? -1 : 0;
        }
    }
}

// ---------------------------------------------------------------- encode / decode / group ops

__device__ __forceinline__ void store_bytes_from_words(uint8_t *dst, const uint32_t *w, int nbytes) {
    for (int i = 0; i < nbytes; i++) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}

GD_KERNEL k_point_encode(uint8_t *__restrict__ ser, const uint64_t *__restrict__ pts, uint32_t n, int eddsa) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        pt p = pt_load_abi(pts + 32 * (size_t)i);
        uint32_t w[15];
        if (eddsa) {  // uniform
            pt_encode_eddsa_words(w, p);
            uint8_t *dst = ser + 57 * (size_t)i;
#pragma unroll 1
            for (int k = 0; k < 57; k++) dst[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        } else {
            pt_encode_words(w, p);
            uint32_t *dst = reinterpret_cast<uint32_t *>(ser + 56 * (size_t)i);  // 56*i is 8-byte aligned
#pragma unroll
            for (int k = 0; k < 14; k++) dst[k] = w[k];
        }
    }
}

GD_KERNEL k_point_decode(uint64_t *__restrict__ pts, int32_t *__restrict__ status,
                         const uint8_t *__restrict__ ser, uint32_t n, int eddsa, int allow_identity) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint32_t w[15];
        pt p;
        bool ok;
        if (eddsa) {
            load_bytes_as_words(w, ser + 57 * (size_t)i, 57, 15);
            ok = pt_decode_eddsa_words(p, w);
        } else {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(ser + 56 * (size_t)i);
#pragma unroll
            for (int k = 0; k < 14; k++) w[k] = src[k];
            ok = pt_decode_words(p, w, allow_identity != 0);
        }
        pt_store_abi(pts + 32 * (size_t)i, p);
        status[i] = ok ? -1 : 0;
    }
}

GD_KERNEL k_point_op(uint64_t *__restrict__ out, const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                     uint32_t n, int op) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        pt p = pt_load_abi(a + 32 * (size_t)i);
        if (op == 2) {
            pt_double(p, true);
        } else {
            pt q = pt_load_abi(b + 32 * (size_t)i);
            p = pt_add(p, q, op == 1);
        }
        pt_store_abi(out + 32 * (size_t)i, p);
    }
}

GD_KERNEL k_point_pred(int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                       const uint64_t *__restrict__ b, uint32_t n, int op) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        pt p = pt_load_abi(a + 32 * (size_t)i);
        bool r;
        if (op == 0) r = pt_eq(p, pt_load_abi(b + 32 * (size_t)i));
        else r = pt_valid(p);
        status[i] = r ? -1 : 0;
    }
}

GD_KERNEL k_field_op(uint64_t *__restrict__ out, int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                     const uint64_t *__restrict__ b, uint32_t n, int op) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        fe x = fe_load_abi(a + 8 * (size_t)i), r;
        bool ok = true;
        if (op == 0) r = fe_mul(x, fe_load_abi(b + 8 * (size_t)i));
        else if (op == 1) r = fe_sqr(x);
        else if (op == 2) r = fe_isr(x, &ok);
        else r = fe_strong(x);
        if (op == 3) {  // canonical limbs, no weak pass on store
            uint64_t *dst = out + 8 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 8; k++) dst[k] = (uint64_t)r.v[2 * k] | (uint64_t)r.v[2 * k + 1] << 28;
        } else {
            fe_store_abi(out + 8 * (size_t)i, r);
        }
        if (status) status[i] = ok ? -1 : 0;
    }
}

// ---------------------------------------------------------------- table staging

// Reference-format comb (80 x {a,b,c} canonical 56-bit limbs) -> ours (28-bit limbs, cn = -c).
GD_KERNEL k_import_comb(uint4 *__restrict__ dst, const uint64_t *__restrict__ src, uint32_t ntables) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < 80 * ntables; i += stride) {
        const uint64_t *s = src + 24 * (size_t)i;
        uint64_t l[8];
        uint4 *d = dst + 12 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 8; k++) l[k] = s[k];
        fe_store(d, fe_weak(fe_from_limbs56(l)));
#pragma unroll
        for (int k = 0; k < 8; k++) l[k] = s[8 + k];
        fe_store(d + 4, fe_weak(fe_from_limbs56(l)));
#pragma unroll
        for (int k = 0; k < 8; k++) l[k] = s[16 + k];
        fe_store(d + 8, fe_weak(fe_neg(fe_from_limbs56(l))));
    }
}

// T_i[k] = (2k+1) * 256^i * B as affine niels ((Y-X)/2Z, (Y+X)/2Z, 78164 T/2Z), one entry per lane.
// Launch with exactly BWT_ENTRIES lanes (28 blocks): the comb gather needs full waves.
GD_KERNEL k_build_bwt(uint4 *__restrict__ dst, const uint4 *__restrict__ comb) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint32_t s_comb[COMB_LDS_WORDS];
    stage_comb_lds(s_comb, comb);
    LdsShuffleComb tab{s_comb, threadIdx.x & 63u};
    const uint32_t e = blockIdx.x * BLOCK + threadIdx.x;   // < BWT_ENTRIES by construction
    const uint32_t i = e >> 7, kk = e & 127;
    sc v = sc_zero();
    const uint32_t m = 2 * kk + 1;
    const uint32_t bit = 8 * i, wd = bit >> 5, sh = bit & 31;
#pragma unroll
    for (int w = 0; w < 14; w++) {
        uint32_t x = 0;
        if ((uint32_t)w == wd) x = m << sh;
        if ((uint32_t)w == wd + 1 && sh > 23) x = m >> (32 - sh);
        v.w[w] = x;
    }
    LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_reduce(v)));
    pt p = ladder_comb(bits, tab);
    fe zi = fe_invert(fe_weak(fe_add(p.z, p.z)));
    uint4 *q = dst + 12 * (size_t)e;
    fe_store(q, fe_mul(fe_weak(fe_sub<2>(p.y, p.x)), zi));
    fe_store(q + 4, fe_mul(fe_weak(fe_add(p.x, p.y)), zi));
    fe_store(q + 8, fe_mul(fe_mulw(p.t, TWO_EFF_D), zi));
}

// 16-entry window table (our pniels form) of one point, by lane 0
GD_KERNEL k_build_shared_table(uint4 *__restrict__ dst, const uint64_t *__restrict__ point) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        LaneTable t{dst};
        build_window_table(t, pt_load_abi(point));
    }
}

// ref: goldilocks_448_precompute (src/goldilocks.c:755-818).  One table per lane.
// work: PRECOMP_U4 per lane of HBM workspace: 80 x {Y-X, Y+X, T, 2Z, prefix product} + 4 teeth.
GD_KERNEL k_precompute(uint64_t *__restrict__ tables, const uint64_t *__restrict__ base, uint32_t n,
                       uint4 *__restrict__ workspace) {
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    uint4 *work = workspace + (size_t)lane * PRECOMP_U4;
    for (uint32_t i = lane; i < n; i += stride) {
        pt working = pt_load_abi(base + 32 * (size_t)i);
        // entry idx of comb j = sum_k (+-) 2^(18(k+5j)) B, tooth 4 always +, tooth k<4 + iff bit k of idx
#pragma unroll 1
        for (int j = 0; j < 5; j++) {
            // teeth of this comb, kept as doubled pniels for the Gray-code walk
            pt start = working;
            uint4 *teeth = work + 80 * 20;  // 4 pniels behind the 80 entries
#pragma unroll 1
            for (int k = 0; k < 5; k++) {
                if (k) start = pt_add(start, working, false);
                if (k == 4 && j == 4) break;
                pt_double(working, true);
                if (k < 4) LaneTable{teeth}.store(k, pt_to_pniels(working));  // 2 * tooth_k
#pragma unroll 1
                for (int d = 0; d < 17; d++) pt_double(working, d == 16);
            }
#pragma unroll 1
            for (uint32_t g = 0;; g++) {
                const uint32_t gray = g ^ (g >> 1);
                const uint32_t idx = (((j + 1) << 4) - 1) ^ gray;
                uint4 *w = work + (size_t)idx * 20;
                fe_store(w, fe_weak(fe_sub<2>(start.y, start.x)));
                fe_store(w + 4, fe_weak(fe_add(start.x, start.y)));
                fe_store(w + 8, start.t);
                fe_store(w + 12, fe_weak(fe_add(start.z, start.z)));
                if (g >= 15) break;
                const uint32_t delta = (g + 1) ^ ((g + 1) >> 1) ^ gray;  // the Gray bit that flips
                const uint32_t k = 31 - __clz(delta);
                pniels step = LaneTable{teeth}.load(k);
                pt_add_pniels(start, step, /*neg=*/(gray & (1u << k)) == 0, true);
            }
        }
        // Montgomery's trick over the 80 values 2Z (src/goldilocks.c:703-726)
        fe acc = fe_one();
#pragma unroll 1
        for (int e = 0; e < 80; e++) {
            fe_store(work + (size_t)e * 20 + 16, acc);
            acc = fe_mul(acc, fe_load(work + (size_t)e * 20 + 12));
        }
        fe inv = fe_invert(acc);
        uint64_t *dst = tables + (size_t)i * (80 * 24);
#pragma unroll 1
        for (int e = 79; e >= 0; e--) {
            uint4 *w = work + (size_t)e * 20;
            fe zi = fe_mul(inv, fe_load(w + 16));
            inv = fe_mul(inv, fe_load(w + 12));
            fe a = fe_strong(fe_mul(fe_load(w), zi));
            fe b = fe_strong(fe_mul(fe_load(w + 4), zi));
            // c = 2 d' T / (2Z) = -(78164 T) * zi
            fe c = fe_strong(fe_neg(fe_mul(fe_mulw(fe_load(w + 8), TWO_EFF_D), zi)));
            uint64_t *d = dst + 24 * e;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                d[k] = (uint64_t)a.v[2 * k] | (uint64_t)a.v[2 * k + 1] << 28;
                d[8 + k] = (uint64_t)b.v[2 * k] | (uint64_t)b.v[2 * k + 1] << 28;
                d[16 + k] = (uint64_t)c.v[2 * k] | (uint64_t)c.v[2 * k + 1] << 28;
            }
        }
    }
}

}  // namespace gd
