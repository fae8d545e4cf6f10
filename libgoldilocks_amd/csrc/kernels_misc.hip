// kernels_misc.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "kernels.hpp"
#include "gf28s.hpp"
#include "inv_wave.hpp"

namespace gd {

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

// The rest of the X448 surface ("next" row f3): Montgomery u-coordinates out of Edwards data, 56 bytes each.
//   pts == nullptr: u = y^2 (1 - d y^2) / (1 - y^2) of an Ed448 public key (its 57th byte is not read, a y >= p is
//                   taken as it stands: goldilocks_ed448_convert_public_key_to_x448, src/goldilocks.c:1079-1102)
//   pts != nullptr: (y / x)^2 of a point (goldilocks_448_point_mul_by_ratio_and_encode_like_x448, src/goldilocks.c:1104-1115)
// with 1 / 0 = 0 as gf_invert has it (src/goldilocks.c:69-80).  One inversion per WAVE (inv_wave.hpp): wave-uniform rounds.
GD_KERNEL k_x448_from_edwards(uint8_t *__restrict__ out, const uint8_t *__restrict__ ed, const uint64_t *__restrict__ pts,
                              uint32_t n) {
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    uint32_t *const region = s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i0 = blockIdx.x * BLOCK + threadIdx.x; i0 - (threadIdx.x & 63u) < n; i0 += stride) {
        const bool live = i0 < n;
        const uint32_t i = live ? i0 : n - 1;
        fe num, den;
        if (pts) {
            const pt p = pt_load_abi(pts + 32 * (size_t)i);
            num = p.y;
            den = p.x;
        } else {
            uint32_t w[15];
            load_bytes_as_words(w, ed + 57 * (size_t)i, 56, 14);
            const fe y2 = fe_sqr(fe_unpack_words(w));
            den = fe_weak(fe_sub<2>(fe_one(), y2));                                       // 1 - y^2
            num = fe_mul(y2, fe_weak(fe_add(fe_one(), fe_mulw(y2, NEG_EDWARDS_D))));      // y^2 (1 - d y^2)
        }
        const fe q = fe_mul(num, wave_shared_invert(den, region));
        uint32_t w[14];
        fe_serialize_words(w, pts ? fe_sqr(q) : q);
        if (live) store_bytes_from_words(out + 56 * (size_t)i, w, 56);
    }
}

// Test hook: entries [first, first + count) of the base point's window table as canonical bytes -- a, b, cn of each affine
// niels serialized (3 x 56 bytes) -- for the every-entry check of k_build_bwt against the oracle (tests/test_gpu_every_lane.py).
GD_KERNEL k_bwt_export(uint8_t *__restrict__ out, const uint4 *__restrict__ bwt, uint64_t first, uint32_t count) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) {
        const uint4 *q = bwt + BWT_HEADER_U4 + 12 * (first + i);
        uint32_t *dst = reinterpret_cast<uint32_t *>(out + 168 * (size_t)i);      // 168 i is 8-byte aligned
#pragma unroll 1
        for (int f = 0; f < 3; f++) {
            uint32_t w[14];
            fe_serialize_words(w, fe_load(q + 4 * f));
#pragma unroll
            for (int k = 0; k < 14; k++) dst[14 * f + k] = w[k];
        }
    }
}

// What SHAKE256(sk) gives an Ed448 private key's owner besides the public key: 56 bytes each.
//   as_scalar == 0: the X448 private key, SHAKE256(sk)[0:56] (goldilocks_ed448_convert_private_key_to_x448,
//                   src/eddsa.c:83-95)
//   as_scalar != 0: the secret scalar as a scalar_s -- clamp(SHAKE256(sk)[0:57]) mod q, halved twice for the encode ratio
//                   (goldilocks_ed448_derive_secret_scalar, src/eddsa.c:97-128)
// Secret data: nothing is staged in LDS (the word-granular absorb keeps the sponge in registers).
GD_KERNEL k_ed448_expand_secret(uint8_t *__restrict__ out, const uint8_t *__restrict__ sk, uint32_t n, int as_scalar) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint8_t *sk57 = sk + 57 * (size_t)i;
        Ed448Msg m;
        m.a = sk57; m.alen = 57; m.b = sk57; m.blen = 0; m.msg = sk57; m.msglen = 0;
        m.ctx = sk57; m.ctxlen = 0; m.ph = 0; m.dom = false;
        uint32_t w[29];
        int unused = 0;
        shake256_114(w, m, 57, unused);
        uint32_t *dst = reinterpret_cast<uint32_t *>(out + 56 * (size_t)i);   // 56 i is 8-byte aligned
        if (as_scalar) {
            ed448_clamp_words(w);
            const sc s = sc_halve(sc_halve(sc_decode_long_words<57>(w)));
#pragma unroll
            for (int k = 0; k < 14; k++) dst[k] = s.w[k];
        } else {
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

GD_KERNEL k_point_op(uint64_t *out, const uint64_t *a, const uint64_t *__restrict__ b,
                     uint32_t n, int op) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        pt p = pt_load_abi(a + 32 * (size_t)i);
        if (op == 2) {
            pt_double(p, true);
        } else if (op == 3) {
            p = pt_negate(p);
        } else if (op == 4) {                     // the same point of the quotient group: + the 2-torsion point (ref: src/goldilocks.c:675-683)
            p.x = fe_neg(p.x);
            p.y = fe_neg(p.y);
        } else if (op == 5) {                     // the same point, other projective coordinates (ref: src/goldilocks.c:685-701)
            uint32_t w[14];                       // b: 56 bytes per point, the factor (anything, >= p included; 0 counts as 1)
            load_bytes_as_words(w, reinterpret_cast<const uint8_t *>(b) + 56 * (size_t)i, 56, 14);
            fe f;
            (void)fe_deserialize_words(f, w);
            f = fe_select(f, fe_one(), fe_is_zero(f));
            p.x = fe_mul(p.x, f);
            p.y = fe_mul(p.y, f);
            p.z = fe_mul(p.z, f);
            p.t = fe_mul(p.t, f);
        } else {
            pt q = pt_load_abi(b + 32 * (size_t)i);
            p = pt_add(p, q, op == 1);
        }
        pt_store_abi(out + 32 * (size_t)i, p);
    }
}

// Scalars mod q, one operation per lane (ref: src/scalar.c; the reference's API of point_448.h:100-260):
//   0 add  1 sub  2 mul  3 halve          out, a, b: scalar_s (7 x u64, canonical)
//   4 invert: out = a^(q-2) (0 for a = 0), status = success iff that is not 0      (src/scalar.c:107-166)
//   5 decode: a = 56 bytes per operation, out = their value mod q, status = success iff the value was below q   (:233-250)
//   6 decode_long: a = len bytes per operation (any length, 0 included), out = their value mod q                   (:257-293)
GD_KERNEL k_scalar_op(uint64_t *out, int32_t *__restrict__ status, const uint8_t *a, const uint64_t *b, uint32_t n, int op,
                      uint32_t len) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        sc r;
        if (op <= 4) {
            const sc x = sc_from_abi(reinterpret_cast<const uint64_t *>(a) + 7 * (size_t)i);
            if (op == 0) r = sc_add(x, sc_from_abi(b + 7 * (size_t)i));
            else if (op == 1) r = sc_sub(x, sc_from_abi(b + 7 * (size_t)i));
            else if (op == 2) r = sc_mul(x, sc_from_abi(b + 7 * (size_t)i));
            else if (op == 3) r = sc_halve(x);
            else {
                r = x;                                               // the exponent's leading bit
#pragma unroll 1
                for (int k = 444; k >= 0; k--) {                     // q - 2: public, every lane the same branch
                    r = sc_mul(r, r);
                    const uint32_t word = SC_Q[k >> 5] - (k < 32 ? 2u : 0u);
                    if ((word >> (k & 31)) & 1u) r = sc_mul(r, x);
                }
                uint32_t any = 0;
#pragma unroll
                for (int k = 0; k < 14; k++) any |= r.w[k];
                status[i] = any ? -1 : 0;
            }
        } else {
            const uint32_t nbytes = op == 5 ? 56u : len;
            const uint8_t *src = a + (size_t)nbytes * i;
            // 56-byte pieces from the top: acc = acc * 2^448 + piece (mod q)
            sc two448 = sc_zero();                                   // 2^448 mod q = 4 c  (q = 2^446 - c)
            {
                uint64_t carry = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    carry += k < 7 ? (uint64_t)SC_C[k] << 2 : 0;
                    two448.w[k] = (uint32_t)carry;
                    carry >>= 32;
                }
            }
            r = sc_zero();
            bool below_q = true;
            const uint32_t pieces = (nbytes + 55) / 56;
#pragma unroll 1
            for (uint32_t pc = pieces; pc-- > 0;) {
                sc piece;
#pragma unroll 1
                for (int k = 0; k < 14; k++) {
                    uint32_t w = 0;
                    for (int j = 0; j < 4; j++) {
                        const uint32_t at = 56 * pc + 4 * k + j;
                        w |= at < nbytes ? (uint32_t)src[at] << (8 * j) : 0u;
                    }
                    piece.w[k] = w;
                }
                if (op == 5) {                                      // the range check of scalar_decode
                    int64_t acc = 0;
#pragma unroll
                    for (int k = 0; k < 14; k++) acc = (acc + (int64_t)piece.w[k] - (int64_t)SC_Q[k]) >> 32;
                    below_q = acc != 0;
                }
                r = sc_add(sc_mul(r, two448), sc_reduce(piece));
            }
            if (op == 5) status[i] = below_q ? -1 : 0;
        }
        sc_to_abi(out + 7 * (size_t)i, r);
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

// Field-level operations for the parity tests, one element per lane:
//   0 mul  1 sqr  2 isr (+mask)  3 strong_reduce (canonical limbs, stored raw)
//   4 mulw (w = low 32 bits of b's limb 0; src/arch_ref64/f_impl.c:168-190)
//   5 add, 6 sub: the reference's gf_add / gf_sub = RAW op (+ bias) + weak_reduce (src/field.h:40-55,
//     arch_ref64/f_impl.h:10-38)          7 weak_reduce alone
//   8 eq (mask), 9 lobit (mask)           (src/f_generic.c:107-131)
//   10 serialize: 56 bytes in the first 7 words of out; 11 deserialize: a holds 56 bytes, status =
//      "value < p" mask, out = limbs (src/f_generic.c:19-68)
//   12 mul at magnitudes: (ma*a) * (mb*b) with the limb-wise multiples formed WITHOUT reduction
//      (ma = aux & 0xff, mb = aux >> 8): the worst cases of the magnitude contract in gf28.hpp go
//      through the device build of fe_mul this way;  13 sqr at magnitude ma
// Inputs are loaded WITHOUT a weak pass for ops 5-7 so that unreduced limbs reach the device code as given.
__device__ __forceinline__ fe fe_load_abi_raw(const uint64_t *p) {
    uint64_t l[8];
#pragma unroll
    for (int k = 0; k < 8; k++) l[k] = p[k];
    return fe_from_limbs56(l);
}
__device__ __forceinline__ void fe_store_limbs_raw(uint64_t *dst, const fe &r) {
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = (uint64_t)r.v[2 * k] + ((uint64_t)r.v[2 * k + 1] << 28);
}
GD_KERNEL k_field_op(uint64_t *__restrict__ out, int32_t *__restrict__ status, const uint64_t *__restrict__ a,
                     const uint64_t *__restrict__ b, uint32_t n, int op, uint32_t aux) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint64_t *pa = a + 8 * (size_t)i, *pb = b ? b + 8 * (size_t)i : nullptr;
        uint64_t *dst = out ? out + 8 * (size_t)i : nullptr;
        bool ok = true;
        if (op <= 3) {
            fe x = fe_load_abi(pa), r;
            if (op == 0) r = fe_mul(x, fe_load_abi(pb));
            else if (op == 1) r = fe_sqr(x);
            else if (op == 2) r = fe_isr(x, &ok);
            else r = fe_strong(x);
            if (op == 3) fe_store_limbs_raw(dst, r);   // canonical limbs, no weak pass on store
            else fe_store_abi(dst, r);
        } else if (op == 4) {
            fe_store_abi(dst, fe_mulw(fe_load_abi(pa), (uint32_t)pb[0]));
        } else if (op == 5) {
            fe_store_limbs_raw(dst, fe_weak(fe_add(fe_load_abi_raw(pa), fe_load_abi_raw(pb))));
        } else if (op == 6) {
            fe_store_limbs_raw(dst, fe_weak(fe_sub<2>(fe_load_abi_raw(pa), fe_weak(fe_load_abi_raw(pb)))));
        } else if (op == 7) {
            fe_store_limbs_raw(dst, fe_weak(fe_load_abi_raw(pa)));
        } else if (op == 8) {
            ok = fe_eq(fe_load_abi(pa), fe_load_abi(pb));
        } else if (op == 9) {
            ok = fe_lobit(fe_load_abi(pa));
        } else if (op == 10) {
            uint32_t w[14];
            fe_serialize_words(w, fe_load_abi(pa));
#pragma unroll
            for (int k = 0; k < 7; k++) dst[k] = (uint64_t)w[2 * k] | (uint64_t)w[2 * k + 1] << 32;
            dst[7] = 0;
        } else if (op == 11) {
            uint32_t w[14];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                w[2 * k] = (uint32_t)pa[k];
                w[2 * k + 1] = (uint32_t)(pa[k] >> 32);
            }
            fe x;
            ok = fe_deserialize_words(x, w);
            fe_store_limbs_raw(dst, x);
        } else if (op == 14 || op == 15) {
            // the signed, register-paired layer of the ladders (gf28s.hpp) through the DEVICE build -- its pair-wise
            // additions are inline v_lshl_add_u64 there, which no host build runs.  A multiplicity k: k copies added
            // pair-wise (1 .. 3), or, with bit 7 set, the limb-wise NEGATIVE of that many (a "signed" operand).
            const auto operand = [&](const uint64_t *p, uint32_t k, auto use) {
                const sfp x = sfe_from_fe(fe_load_abi(p));
                const int m = (int)(k & 0x7f);
                if (k & 0x80) {
                    sfs xs = sfe_sub(sfe_from_fe(fe_zero()), x);
                    for (int t = 1; t < m; t++) xs = sfe_sub(xs, x);
                    use(xs);
                } else {
                    sfp xs = x;
                    for (int t = 1; t < m; t++) xs = sfe_add(xs, x);
                    use(xs);
                }
            };
            const uint32_t ka = aux & 0xff, kb = aux >> 8;
            if (op == 14) {
                operand(pa, ka, [&](const auto &xs) {
                    operand(pb, kb, [&](const auto &ys) { fe_store_abi(dst, sfe_to_fe(sfe_mul(xs, ys))); });
                });
            } else if (kb) {   // the square of a sum of products (columns 0 - 2 of the high half read as unsigned)
                const sfp x = sfe_from_fe(fe_load_abi(pa));
                sfp xs = x;
                for (int t = 1; t < (int)(ka & 0x7f); t++) xs = sfe_add(xs, x);
                fe_store_abi(dst, sfe_to_fe(sfe_sqr<true>(xs)));
            } else {
                operand(pa, ka, [&](const auto &xs) { fe_store_abi(dst, sfe_to_fe(sfe_sqr<false>(xs))); });
            }
        } else {
            const int ma = (int)(aux & 0xff), mb = (int)(aux >> 8);
            fe x = fe_load_abi(pa), xs = fe_zero();
            for (int k = 0; k < ma; k++) xs = fe_add(xs, x);
            if (op == 12) {
                fe y = fe_load_abi(pb), ys = fe_zero();
                for (int k = 0; k < mb; k++) ys = fe_add(ys, y);
                fe_store_abi(dst, fe_mul(xs, ys));
            } else {
                fe_store_abi(dst, fe_sqr(xs));
            }
        }
        if (status) status[i] = ok ? -1 : 0;
    }
}

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

}  // namespace gd
