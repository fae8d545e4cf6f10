// kernels_wave.hip -- one operation per wavefront: the small-batch / single-call kernels (wave_coop.hpp).
#include "kernels.hpp"
#include "wave_coop.hpp"

namespace gd {

// scaled[i] = scalar[i] * base[i], one operation per wave   (ref: goldilocks_448_point_scalarmul)
// out may alias base.  Index-independent table access whatever the table mode (the LDS scan is cheap).
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_scalarmul_wave(uint64_t *out, const uint64_t *base,
                                                                           const uint64_t *__restrict__ scalar,
                                                                           uint32_t n) {
    __shared__ uint32_t s_tab[BLOCK / 64][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab{s_tab[w]};
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        const wc::wfe B = wc::load_point(L, base + 32 * (size_t)op);
        const sc k = sc_load_abi(scalar + 7 * (size_t)op);
        const wc::wfe r = wc::scalarmul(L, tab, s_bits[w], B, k);
        wc::store_point(L, out + 32 * (size_t)op, r);
    }
    // the recoded scalar does not stay behind in LDS (the table holds multiples of the caller's point)
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
}

// ser[i] = the decaf (56 bytes) or RFC 8032 (57 bytes, 4 * pts[i]) encoding of pts[i], one point per wave
// (ref: goldilocks_448_point_encode, goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa)
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_encode_wave(uint8_t *__restrict__ ser, const uint64_t *__restrict__ pts,
                                                                        uint32_t n, int eddsa) {
    const wc::Lane L = wc::make_lane();
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); op < n; op += nwaves) {   // wave-uniform
        const wc::wfe P = wc::load_point(L, pts + 32 * (size_t)op);
        if (eddsa) wc::encode_eddsa(L, ser + 57 * (size_t)op, P);   // uniform
        else wc::encode(L, ser + 56 * (size_t)op, P);
    }
}
// pts[i] = the point ser[i] encodes, status[i] = it decodes; one encoding per wave
// (ref: goldilocks_448_point_decode, goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio)
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_decode_wave(uint64_t *__restrict__ pts, int32_t *__restrict__ status,
                                                                        const uint8_t *__restrict__ ser, uint32_t n, int eddsa,
                                                                        int allow_identity) {
    const wc::Lane L = wc::make_lane();
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); op < n; op += nwaves) {   // wave-uniform
        wc::wfe P;
        bool ok;
        if (eddsa) {   // uniform; every row decodes the same 57 bytes
            wc::wfe X, Y, Z, T;
            const bool okrow = wc::decode_eddsa_rows(L, ser + 57 * (size_t)op, X, Y, Z, T);
            ok = (__builtin_amdgcn_ballot_w64(okrow) & 1u) != 0;
            P = wc::pack_point<0>(L, X, Y, Z, T);
        } else {
            ok = wc::decode(L, ser + 56 * (size_t)op, allow_identity != 0, P);
        }
        wc::store_point(L, pts + 32 * (size_t)op, P);
        if ((threadIdx.x & 63u) == 0) status[op] = ok ? -1 : 0;
    }
}

// out[i] = Elligator 2 of hash[i] (56 bytes), or the sum of the maps of its two halves (112 bytes, uniform): one
// output per wave, the two halves in rows 0 and 1 of the same instruction stream
// (ref: goldilocks_448_point_from_hash_nonuniform / _uniform, src/elligator.c:32-94)
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_from_hash_wave(uint64_t *__restrict__ out, const uint8_t *__restrict__ hash,
                                                                           uint32_t n, int uniform) {
    const wc::Lane L = wc::make_lane();
    const uint32_t nwaves = gridDim.x * (BLOCK / 64), nb = uniform ? 112 : 56;
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); op < n; op += nwaves) {   // wave-uniform
        const uint8_t *str = hash + (size_t)nb * op + (uniform && (L.row & 1u) ? 56 : 0);
        wc::wfe X, Y, Z, T;
        wc::from_hash_rows(L, str, X, Y, Z, T);
        wc::wfe P = wc::pack_point<0>(L, X, Y, Z, T);
        if (uniform) {
            const uint32_t swap_row = L.row ^ 1u;
            P = wc::add_entry(L, P, wc::to_pniels(L, wc::pack_point<1>(L, X, Y, Z, T), swap_row), false, swap_row);
        }
        wc::store_point(L, out + 32 * (size_t)op, P);
    }
}

// tables[i] = the 5 x 5 x 18 comb of base[i] in the reference's format, one table per wave (ref: goldilocks_448_precompute)
extern "C" __global__ void __launch_bounds__(BLOCK) k_precompute_wave(uint64_t *__restrict__ tables, const uint64_t *__restrict__ base,
                                                                      uint32_t n) {
    __shared__ uint32_t s_work[BLOCK / 64][wc::PRECOMP_WAVE_WORDS];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves)   // wave-uniform
        wc::precompute(L, s_work[w], tables + (size_t)op * (80 * 24), wc::load_point(L, base + 32 * (size_t)op));
}

// Verification keys that get a comb of their own (kernels_verify.hip): key k decoded and its 28 teeth 2^(16 m) * A_k
// (and their doubles, which the comb's entries are walked with) written as pniels, ONE KEY PER WAVE -- the chain of 432 doublings is pure latency, and a wave's doubling is two row
// multiplications (0.25 ms for the chain instead of the 1.7 ms of a lane's).  A pniels' 64 words are the wave's 64
// lanes (rows a, b, cn, z): every tooth is one coalesced 256-byte store.
extern "C" __global__ void __launch_bounds__(BLOCK) k_verify_key_teeth(uint4 *__restrict__ teeth, uint8_t *__restrict__ key_ok,
                                                                       const uint32_t *__restrict__ ctrl,
                                                                       const uint32_t *__restrict__ key_list,
                                                                       const uint8_t *__restrict__ pk) {
    const wc::Lane L = wc::make_lane();
    const uint32_t combed = ctrl[2], me = threadIdx.x & 63u, swap_row = L.row ^ 1u;
    if (combed > (uint32_t)KEY_TEETH_BY_WAVE_MAX) return;   // many keys: a lane each (k_verify_key_teeth_lanes)
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t k = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); k < combed; k += nwaves) {   // wave-uniform
        wc::wfe X, Y, Z, T;
        const bool ok = wc::decode_eddsa_rows(L, pk + 57 * (size_t)key_list[k], X, Y, Z, T);
        if (me == 0) key_ok[k] = ok ? 1 : 0;
        wc::wfe P = wc::pack_point<0>(L, X, Y, Z, T);
        uint32_t *out = reinterpret_cast<uint32_t *>(teeth + (size_t)KEY_TEETH_U4 * k);
        const int NT = (int)(key_comb_combs(ctrl[3]) * ctrl[3]), spacing = (int)key_comb_spacing(ctrl[3]);   // 4 x 7 teeth, spacing 16; 4 x 8, 14; 5 x 9, 10 (k_verify_key_mode)
#pragma unroll 1
        for (int m = 0; m < NT; m++) {          // T_m, and 2 T_m (the first doubling towards T_(m+1)) behind the NT teeth
            out[64 * m + me] = wc::to_pniels(L, P, swap_row);
            P = wc::dbl(L, P);
            out[64 * (NT + m) + me] = wc::to_pniels(L, P, swap_row);
            if (m + 1 == NT) break;
#pragma unroll 1
            for (int d = 1; d < spacing; d++) P = wc::dbl(L, P);
        }
    }
}

// scaled[i] = encode(scalar[i] * decode(base[i])), one operation per wave; an encoding that does not decode gives
// status 0 and (unless short_circuit) the base point is multiplied instead   (ref: goldilocks_448_direct_scalarmul,
// src/goldilocks.c:888-903)
extern "C" __global__ void __launch_bounds__(BLOCK) k_direct_scalarmul_wave(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                                                                            const uint8_t *__restrict__ base,
                                                                            const uint64_t *__restrict__ scalar, uint32_t n,
                                                                            int allow_identity, int short_circuit,
                                                                            const uint64_t *__restrict__ point_base_abi) {
    __shared__ uint32_t s_tab[BLOCK / 64][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab{s_tab[w]};
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        wc::wfe P;
        const bool ok = wc::decode(L, base + 56 * (size_t)op, allow_identity != 0, P);
        if ((threadIdx.x & 63u) == 0) status[op] = ok ? -1 : 0;
        if (!ok && short_circuit) continue;                 // the encoding is public: so is this branch
        if (!ok) P = wc::load_point(L, point_base_abi);     // src/goldilocks.c:898
        const wc::wfe r = wc::scalarmul(L, tab, s_bits[w], P, sc_load_abi(scalar + 7 * (size_t)op));
        wc::encode(L, scaled + 56 * (size_t)op, r);
    }
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
}

// (a1[i], a2[i]) = (s1[i], s2[i]) * base[i], one operation per wave: the table is built once and walked twice
// (ref: goldilocks_448_point_dual_scalarmul, src/goldilocks.c:543-642).  a1 may alias base.
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_dual_scalarmul_wave(uint64_t *a1, uint64_t *a2, const uint64_t *base,
                                                                                const uint64_t *__restrict__ s1,
                                                                                const uint64_t *__restrict__ s2, uint32_t n) {
    __shared__ uint32_t s_tab[BLOCK / 64][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab{s_tab[w]};
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        wc::build_table(L, tab, wc::load_point(L, base + 32 * (size_t)op));
        const wc::wfe r1 = wc::walk_table(L, tab, s_bits[w], sc_load_abi(s1 + 7 * (size_t)op));
        const wc::wfe r2 = wc::walk_table(L, tab, s_bits[w], sc_load_abi(s2 + 7 * (size_t)op));
        wc::store_point(L, a1 + 32 * (size_t)op, r1);
        wc::store_point(L, a2 + 32 * (size_t)op, r2);
    }
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
}

// combo[i] = s1[i]*b1[i] + s2[i]*b2[i], one operation per wave; b1 == nullptr: b1 is the base point through
// its window table (goldilocks_448_base_double_scalarmul_non_secret: public scalars by contract).
// (ref: goldilocks_448_point_double_scalarmul, src/goldilocks.c:467-541, :1260-1330).  out may alias b2.
extern "C" __global__ void __launch_bounds__(BLOCK) k_double_scalarmul_wave(uint64_t *out, const uint64_t *b1,
                                                                            const uint64_t *__restrict__ s1, const uint64_t *b2,
                                                                            const uint64_t *__restrict__ s2, uint32_t n,
                                                                            const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_tab[BLOCK / 64][2][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][32];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab{s_tab[w][0]}, tab1{s_tab[w][1]};
    uint32_t *bits = s_bits[w];
    struct Bits {
        const uint32_t *p;
        __device__ __forceinline__ uint32_t word(int k) const { return p[k]; }
    };
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        const sc k1 = sc_load_abi(s1 + 7 * (size_t)op), k2 = sc_load_abi(s2 + 7 * (size_t)op);
        wc::wfe P;
        if (b1) {   // uniform: two caller points share one 90-window ladder
            wc::build_table(L, tab, wc::load_point(L, b2 + 32 * (size_t)op));
            wc::build_table(L, tab1, wc::load_point(L, b1 + 32 * (size_t)op));
            const sc r1 = sc_recode_signed(k1), r2 = sc_recode_signed(k2);
#pragma unroll
            for (int k = 0; k < 14; k++) {
                bits[k] = r1.w[k];
                bits[16 + k] = r2.w[k];
            }
            bits[14] = bits[30] = 0;
            P = wc::walk_two_tables(L, tab1, tab, Bits{bits}, Bits{bits + 16}, 90, false, false);
        } else {
            P = wc::scalarmul(L, tab, bits, wc::load_point(L, b2 + 32 * (size_t)op), k2);
            const sc r = sc_recode_bwt(k1, GlobalBwt{bwt});
#pragma unroll
            for (int k = 0; k < 14; k++) bits[k] = r.w[k];
            bits[14] = 0;
            P = wc::add_base_multiple(L, P, Bits{bits}, bwt);
        }
        wc::store_point(L, out + 32 * (size_t)op, P);
    }
    if ((threadIdx.x & 63u) < 32) s_bits[w][threadIdx.x & 31u] = 0;
}

// shared[i] = X448(scalar[i], base[i]), one ladder per wave   (ref: goldilocks_x448, src/goldilocks.c:1006-1076)
extern "C" __global__ void __launch_bounds__(BLOCK) k_x448_wave(uint8_t *__restrict__ shared, int32_t *__restrict__ status,
                                                                const uint8_t *__restrict__ base,
                                                                const uint8_t *__restrict__ scalar, uint32_t n) {
    const wc::Lane L = wc::make_lane();
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); op < n; op += nwaves) {   // wave-uniform
        uint32_t k[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(scalar + 56 * (size_t)op);
#pragma unroll
        for (int j = 0; j < 14; j++) k[j] = src[j];
        const bool ok = wc::x448(L, shared + 56 * (size_t)op, base + 56 * (size_t)op, k);
        if (status && (threadIdx.x & 63u) == 0) status[op] = ok ? -1 : 0;
    }
}

// scaled[i] = scalar[i] * G, G given by a comb table in this library's form   (ref: goldilocks_448_precomputed_scalarmul)
extern "C" __global__ void __launch_bounds__(BLOCK) k_precomputed_scalarmul_wave(uint64_t *__restrict__ out,
                                                                                 const uint4 *__restrict__ comb,
                                                                                 const uint64_t *__restrict__ scalar, uint32_t n) {
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6, nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        const sc k = sc_load_abi(scalar + 7 * (size_t)op);
        wc::store_point(L, out + 32 * (size_t)op, wc::comb_scalarmul(L, comb, wc::put_bits(s_bits[w], sc_recode_signed(k))));
    }
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
}

// pk[i] = derive_public_key(sk[i]) / out[i] = x448_derive_public_key(scalar[i]), one per wave
extern "C" __global__ void __launch_bounds__(BLOCK) k_derive_wave(uint8_t *__restrict__ out, const uint8_t *__restrict__ sk,
                                                                  uint32_t n, const uint4 *__restrict__ comb, int x448_keygen) {
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6, nwaves = gridDim.x * (BLOCK / 64);
    LdsStage stage{s_stage + threadIdx.x};
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        if (x448_keygen) {   // uniform
            uint32_t k[14];
            const uint32_t *src = reinterpret_cast<const uint32_t *>(sk + 56 * (size_t)op);
#pragma unroll
            for (int j = 0; j < 14; j++) k[j] = src[j];
            const wc::wfe P = wc::comb_scalarmul(L, comb, wc::put_bits(s_bits[w], sc_recode_signed(x448_public_scalar(k))));
            wc::encode_x448(L, out + 56 * (size_t)op, P);
        } else {
            wc::derive(L, out + 57 * (size_t)op, sk + 57 * (size_t)op, comb, s_bits[w], stage);
        }
    }
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
    lds_wipe_lane(s_stage + threadIdx.x, 34);
}

// sig[i] = sign(sk[i], pk[i], msg[i]), one signature per wave   (ref: goldilocks_ed448_sign)
extern "C" __global__ void __launch_bounds__(BLOCK) k_ed448_sign_wave(
    uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk, const uint8_t *__restrict__ msgs,
    const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx,
    uint32_t ctx_len, uint32_t n, const uint4 *__restrict__ comb) {
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    __shared__ uint32_t s_stage[34 * BLOCK];
    __shared__ uint8_t s_bytes[BLOCK / 64][128];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6, nwaves = gridDim.x * (BLOCK / 64);
    LdsStage stage{s_stage + threadIdx.x};
    for (uint32_t i = blockIdx.x * (BLOCK / 64) + w; i < n; i += nwaves) {   // wave-uniform
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;
        wc::sign(L, sig + 114 * (size_t)i, sk + 57 * (size_t)i, pk + 57 * (size_t)i, msg, fits ? (uint32_t)len64 : 0u, prehashed,
                 ctx, ctx_len, comb, s_bits[w], s_bytes[w], stage);
        if (!fits && (threadIdx.x & 63u) < 57) {
            sig[114 * (size_t)i + (threadIdx.x & 63u)] = 0;
            sig[114 * (size_t)i + 57 + (threadIdx.x & 63u)] = 0;
        }
    }
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
    lds_wipe_lane(s_stage + threadIdx.x, 34);
}

// Field-level test hook for the row arithmetic: every wave takes FOUR consecutive elements (one per row).
//   0 mul  1 strong_reduce (canonical limbs)  2 isr (+ mask)  3 eq (mask)  4 lobit (mask)
//   5 deserialize: a holds 56 bytes; out = limbs, status = value < p
extern "C" __global__ void __launch_bounds__(BLOCK) k_wave_field_op(uint64_t *__restrict__ out, int32_t *__restrict__ status,
                                                                    const uint64_t *__restrict__ a,
                                                                    const uint64_t *__restrict__ b, uint32_t n, int op) {
    const wc::Lane L = wc::make_lane();
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    for (uint32_t g = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); 4 * g < n; g += nwaves) {
        const uint32_t e = 4 * g + L.row;
        const bool live = e < n;
        const size_t at = 8 * (size_t)(live ? e : n - 1);
        auto ld = [&](const uint64_t *p) {
            const uint64_t l56 = p[at + (L.i >> 1)];
            return wc::weak(L, (L.i & 1u) ? (uint32_t)(l56 >> 28) : (uint32_t)l56 & M28);
        };
        bool ok = true;
        wc::wfe r = 0;
        if (op == 0) r = wc::mul(L, ld(a), ld(b));
        else if (op == 1) r = wc::strong(L, ld(a));
        else if (op == 2) r = wc::isr(L, ld(a), ok);
        else if (op == 3) ok = wc::eq(L, ld(a), ld(b));
        else if (op == 4) ok = wc::lobit(L, ld(a));
        else if (op == 5) r = wc::deserialize(L, reinterpret_cast<const uint8_t *>(a + at), ok);
        else {   // 6..9: coordinate X, Y, Z, T of the EdDSA decoding of the 57 bytes at a
            wc::wfe X, Y, Z, T;
            ok = wc::decode_eddsa_rows(L, reinterpret_cast<const uint8_t *>(a + at), X, Y, Z, T);
            r = op == 6 ? X : op == 7 ? Y : op == 8 ? Z : T;
        }
        if (op != 1 && op != 5) r = wc::weak(L, r);
        const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x101, 0xF, 0xF, false);
        if (live && out && !(L.i & 1u)) out[at + (L.i >> 1)] = (uint64_t)r + ((uint64_t)nb << 28);
        if (live && status && L.i == 0) status[e] = ok ? -1 : 0;
    }
}

// status[i] = ed448_verify(sig[i], pk[i], msg[i]), one verification per wave   (ref: goldilocks_ed448_verify)
extern "C" __global__ void __launch_bounds__(BLOCK) k_ed448_verify_wave(
    int32_t *__restrict__ status, const uint8_t *__restrict__ sig, const uint8_t *__restrict__ pk,
    const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets, uint32_t msg_len, uint32_t prehashed,
    const uint8_t *__restrict__ ctx, uint32_t ctx_len, uint32_t n, const uint4 *__restrict__ bwt) {
    __shared__ uint32_t s_tab[BLOCK / 64][2][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][32];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab_a{s_tab[w][0]}, tab_r{s_tab[w][1]};
    LdsStage stage{s_stage + threadIdx.x};
    for (uint32_t i = blockIdx.x * (BLOCK / 64) + w; i < n; i += nwaves) {   // wave-uniform
        const uint8_t *msg = msg_offsets ? msgs + msg_offsets[i] : msgs + (size_t)msg_len * i;
        const uint64_t len64 = msg_offsets ? msg_offsets[i + 1] - msg_offsets[i] : (uint64_t)msg_len;
        const bool fits = len64 < MAX_MESSAGE_BYTES;
        const uint32_t mlen = fits ? (uint32_t)len64 : 0u;
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msg, mlen, prehashed, ctx, ctx_len);
        const bool ok = wc::verify(L, tab_a, tab_r, s_bits[w], m, stage, bwt);
        if ((threadIdx.x & 63u) == 0) status[i] = ok && fits ? -1 : 0;
    }
}

}  // namespace gd
