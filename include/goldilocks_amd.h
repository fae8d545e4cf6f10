/*
 * goldilocks_amd.h -- C ABI of libgoldilocks_amd.so: the MI355X (gfx950) backend for
 * the batched Ed448-Goldilocks scalar-multiplication hot path of otrv4/libgoldilocks.
 *
 * Three groups of entry points, all `extern "C"`, plain pointers and sizes:
 *
 *  (1) DROP-IN single-operation functions with the reference's exact names, argument
 *      meaning and error behaviour, so that eddsa.c / elligator.c / the C++ wrappers /
 *      python/edgold link against this library instead of the arch_* CPU backend.
 *      Each one runs the HIP path with a batch of one (there is NO CPU fallback: if no
 *      gfx950 device can be opened the call aborts with a message on stderr, because
 *      the reference's void functions have no way to report failure).
 *
 *  (2) *_batch functions over HOST arrays (AoS, the reference's struct layouts): one
 *      H2D copy, one kernel, one D2H copy.  New; the reference has no batch API.
 *
 *  (3) goldilocks_amd_*_dev functions over DEVICE arrays on a caller-supplied HIP
 *      stream (what bench.py times; what a host written in another language binds
 *      through cgo/JNI/ctypes -- see INTEGRATION.md).
 *
 * "ref:" comments cite the reference declaration each function replaces, relative to
 * the reference tree (src/public_include/goldilocks/...).
 *
 * Type layouts are the reference's (ref: point_448.h:33-35, 66-70, 82-86): a field
 * element is 8 x uint64 limbs of 56 bits, 32-byte aligned; a point is {x,y,z,t}; a
 * scalar is 7 x uint64 little-endian, fully reduced mod q.  Points produced by this
 * library are weakly reduced like the reference's (limbs < 2^56 + small); as in the
 * reference, raw limbs are not a canonical form -- compare encodings.
 */
#ifndef GOLDILOCKS_AMD_H
#define GOLDILOCKS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOLDILOCKS_AMD_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ ABI types */

/* This header can be included next to the reference's own <goldilocks.h> (include that one FIRST):
 * every type, macro and inline function the reference already declares is skipped here, and the
 * prototypes below are then plain re-declarations of the reference's (same types). */
#ifndef __GOLDILOCKS_COMMON_H__
typedef uint64_t goldilocks_word_t;  /* ref: common.h:60 */
typedef uint64_t goldilocks_bool_t;  /* all-ones / zero mask; ref: common.h:62 */
typedef enum {                       /* ref: common.h:82-85 */
    GOLDILOCKS_SUCCESS = -1,
    GOLDILOCKS_FAILURE = 0
} goldilocks_error_t;
#endif

#ifndef __GOLDILOCKS_ED448_H__
#define GOLDILOCKS_EDDSA_448_PUBLIC_BYTES 57   /* ref: ed448.h:24 */
#define GOLDILOCKS_EDDSA_448_PRIVATE_BYTES 57  /* ref: ed448.h:27 */
#define GOLDILOCKS_EDDSA_448_SIGNATURE_BYTES 114 /* ref: ed448.h:30 */
#endif

#ifndef __GOLDILOCKS_POINT_448_H__
#define GOLDILOCKS_448_SER_BYTES 56            /* ref: point_448.h:40 */
#define GOLDILOCKS_448_SCALAR_BYTES 56         /* ref: point_448.h:48 */
#define GOLDILOCKS_448_SCALAR_LIMBS 7          /* ref: point_448.h:23 */
#define GOLDILOCKS_X448_PUBLIC_BYTES 56        /* ref: point_448.h:60 */
#define GOLDILOCKS_X448_PRIVATE_BYTES 56       /* ref: point_448.h:63 */

#ifndef __GOLDILOCKS_448_GF_DEFINED__
#define __GOLDILOCKS_448_GF_DEFINED__ 1
typedef struct gf_448_s {            /* ref: point_448.h:33-35 */
    goldilocks_word_t limb[8];
} __attribute__((aligned(32))) gf_448_s, gf_448_p[1];
#endif

typedef struct goldilocks_448_point_s {   /* ref: point_448.h:66-70 */
    gf_448_p x, y, z, t;
} goldilocks_448_point_s, goldilocks_448_point_p[1];

typedef struct goldilocks_448_scalar_s {  /* ref: point_448.h:82-86 */
    goldilocks_word_t limb[GOLDILOCKS_448_SCALAR_LIMBS];
} goldilocks_448_scalar_s, goldilocks_448_scalar_p[1];

/* Opaque, caller-allocated, goldilocks_448_sizeof_precomputed_s bytes (15360),
 * goldilocks_448_alignof_precomputed_s alignment (32: what the reference's widest build exports;
 * this library itself needs 16).  ref: point_448.h:73-79.
 * Contents are bit-compatible with the reference's (80 affine niels, canonical limbs). */
struct goldilocks_448_precomputed_s;
typedef struct goldilocks_448_precomputed_s goldilocks_448_precomputed_s;

/* ref: point_448.h:273-278 */
static inline void goldilocks_448_point_copy(goldilocks_448_point_p a, const goldilocks_448_point_p b) { *a = *b; }
/* ref: point_448.h:218-223 */
static inline void goldilocks_448_scalar_copy(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a) { *out = *a; }
#else
/* the reference names the scalar struct only by its tag; the batch prototypes below use this name */
typedef struct goldilocks_448_scalar_s goldilocks_448_scalar_s;
#endif /* __GOLDILOCKS_POINT_448_H__ */

/* ------------------------------------------------------------------ exported constants */

GOLDILOCKS_AMD_API extern const size_t goldilocks_448_sizeof_precomputed_s;   /* ref: point_448.h:79 */
GOLDILOCKS_AMD_API extern const size_t goldilocks_448_alignof_precomputed_s;  /* ref: point_448.h:79 */
GOLDILOCKS_AMD_API extern const goldilocks_448_scalar_p goldilocks_448_scalar_one;   /* ref: point_448.h:89 */
GOLDILOCKS_AMD_API extern const goldilocks_448_scalar_p goldilocks_448_scalar_zero;  /* ref: point_448.h:92 */
GOLDILOCKS_AMD_API extern const goldilocks_448_point_p goldilocks_448_point_identity; /* ref: point_448.h:95 */
GOLDILOCKS_AMD_API extern const goldilocks_448_point_p goldilocks_448_point_base;     /* ref: point_448.h:98 */
GOLDILOCKS_AMD_API extern const goldilocks_448_precomputed_s *goldilocks_448_precomputed_base; /* ref: point_448.h:101 */

/* ------------------------------------------------------------------ (1) drop-in single ops */

/* Scalars modulo the group order q (ref: point_448.h:100-260, :565-580, :722-729; src/scalar.c).  Arguments are canonical
 * (below q), results are; outputs may alias inputs.  The arithmetic runs on the device like everything else, one launch per
 * call (k_scalar_op); encode, eq, set_unsigned, cond_sel, copy and destroy touch memory only.
 *   decode: GOLDILOCKS_SUCCESS iff the 56 bytes are below q; out = their value mod q either way   (src/scalar.c:233-250)
 *   decode_long: the value of ser_len little-endian bytes (any length, 0 included) mod q            (src/scalar.c:257-293)
 *   invert: out = 1/a; GOLDILOCKS_FAILURE (and out = 0) for a = 0                                   (src/scalar.c:107-166)
 *   halve: out = a/2                                                                                (src/scalar.c:316-332) */
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_448_scalar_decode(goldilocks_448_scalar_p out,
        const unsigned char ser[GOLDILOCKS_448_SCALAR_BYTES]);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_decode_long(goldilocks_448_scalar_p out, const unsigned char *ser,
        size_t ser_len);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_encode(unsigned char ser[GOLDILOCKS_448_SCALAR_BYTES],
        const goldilocks_448_scalar_p s);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_add(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a,
        const goldilocks_448_scalar_p b);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_sub(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a,
        const goldilocks_448_scalar_p b);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_mul(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a,
        const goldilocks_448_scalar_p b);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_halve(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a);
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_448_scalar_invert(goldilocks_448_scalar_p out,
        const goldilocks_448_scalar_p a);
GOLDILOCKS_AMD_API goldilocks_bool_t goldilocks_448_scalar_eq(const goldilocks_448_scalar_p a,
        const goldilocks_448_scalar_p b);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_set_unsigned(goldilocks_448_scalar_p out, uint64_t a);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_cond_sel(goldilocks_448_scalar_p out, const goldilocks_448_scalar_p a,
        const goldilocks_448_scalar_p b, goldilocks_word_t pick_b);
GOLDILOCKS_AMD_API void goldilocks_448_scalar_destroy(goldilocks_448_scalar_p scalar);

/* scaled = scalar * base.  Output may alias input.  ref: point_448.h:355-359, src/goldilocks.c:405-465 */
GOLDILOCKS_AMD_API void goldilocks_448_point_scalarmul(goldilocks_448_point_p scaled,
        const goldilocks_448_point_p base, const goldilocks_448_scalar_p scalar);

/* Decode base, multiply, encode.  On a decoding failure: with short_circuit the error is
 * returned and `scaled` is untouched, otherwise the multiply runs on the base point and the
 * error is returned afterwards.  ref: point_448.h:378-384, src/goldilocks.c:888-903 */
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_448_direct_scalarmul(uint8_t scaled[GOLDILOCKS_448_SER_BYTES],
        const uint8_t base[GOLDILOCKS_448_SER_BYTES], const goldilocks_448_scalar_p scalar,
        goldilocks_bool_t allow_identity, goldilocks_bool_t short_circuit);

/* Build the 5x5x18 comb table for `base`.  ref: point_448.h:461-464, src/goldilocks.c:755-818 */
GOLDILOCKS_AMD_API void goldilocks_448_precompute(goldilocks_448_precomputed_s *table,
        const goldilocks_448_point_p base);

/* scaled = scalar * (table's point).  ref: point_448.h:477-481, src/goldilocks.c:830-877 */
GOLDILOCKS_AMD_API void goldilocks_448_precomputed_scalarmul(goldilocks_448_point_p scaled,
        const goldilocks_448_precomputed_s *table, const goldilocks_448_scalar_p scalar);

/* combo = scalar1*base1 + scalar2*base2.  ref: point_448.h:496-502, src/goldilocks.c:467-541 */
GOLDILOCKS_AMD_API void goldilocks_448_point_double_scalarmul(goldilocks_448_point_p combo,
        const goldilocks_448_point_p base1, const goldilocks_448_scalar_p scalar1,
        const goldilocks_448_point_p base2, const goldilocks_448_scalar_p scalar2);

/* combo = scalar1*point_base + scalar2*base2 ("non_secret": the reference is variable-time
 * here; this backend runs a lane-uniform ladder).  ref: point_448.h:542-547, src/goldilocks.c:1260-1330 */
GOLDILOCKS_AMD_API void goldilocks_448_base_double_scalarmul_non_secret(goldilocks_448_point_p combo,
        const goldilocks_448_scalar_p scalar1, const goldilocks_448_point_p base2,
        const goldilocks_448_scalar_p scalar2);

/* Canonical 56-byte decaf encoding.  ref: point_448.h:241-244, src/goldilocks.c:136-140 */
GOLDILOCKS_AMD_API void goldilocks_448_point_encode(uint8_t ser[GOLDILOCKS_448_SER_BYTES],
        const goldilocks_448_point_p pt);
/* FAILURE for non-canonical / negative / (unless allowed) identity encodings; output undefined
 * then.  ref: point_448.h:258-264, src/goldilocks.c:142-176 */
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_448_point_decode(goldilocks_448_point_p pt,
        const uint8_t ser[GOLDILOCKS_448_SER_BYTES], goldilocks_bool_t allow_identity);

/* RFC 8032 57-byte encoding of 4*p / decoding (no cofactor multiply on decode: ratio 4/4).
 * ref: ed448.h:213-232, src/goldilocks.c:905-1004 */
GOLDILOCKS_AMD_API void goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa(
        uint8_t enc[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES], const goldilocks_448_point_p p);
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio(
        goldilocks_448_point_p p, const uint8_t enc[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES]);

/* Group law and predicates.  ref: point_448.h:274-337, src/goldilocks.c:178-268, 644-673 */
GOLDILOCKS_AMD_API goldilocks_bool_t goldilocks_448_point_eq(const goldilocks_448_point_p a,
        const goldilocks_448_point_p b);
GOLDILOCKS_AMD_API goldilocks_bool_t goldilocks_448_point_valid(const goldilocks_448_point_p a);
GOLDILOCKS_AMD_API void goldilocks_448_point_add(goldilocks_448_point_p sum,
        const goldilocks_448_point_p a, const goldilocks_448_point_p b);
GOLDILOCKS_AMD_API void goldilocks_448_point_sub(goldilocks_448_point_p diff,
        const goldilocks_448_point_p a, const goldilocks_448_point_p b);
GOLDILOCKS_AMD_API void goldilocks_448_point_double(goldilocks_448_point_p two_a,
        const goldilocks_448_point_p a);
/* nega = -a.  ref: point_448.h:343-346, src/goldilocks.c:260-268 */
GOLDILOCKS_AMD_API void goldilocks_448_point_negate(goldilocks_448_point_p nega,
        const goldilocks_448_point_p a);
/* The reference's two debugging helpers, which its own test suite (test/test_goldilocks.cxx:379-381, :588, :609) uses to
 * show that a point's OTHER representations -- shifted by the 2-torsion point (-x, -y, z, t), or with all four coordinates
 * scaled by a field element (56 bytes, little-endian, any value; 0 counts as 1) -- encode and behave alike.
 * ref: point_448.h:593-617, src/goldilocks.c:675-701 */
GOLDILOCKS_AMD_API void goldilocks_448_point_debugging_torque(goldilocks_448_point_p q,
        const goldilocks_448_point_p p);
GOLDILOCKS_AMD_API void goldilocks_448_point_debugging_pscale(goldilocks_448_point_p q,
        const goldilocks_448_point_p p, const uint8_t factor[56]);
/* Memory-only helpers of the reference API (no field arithmetic, so nothing to launch):
 * constant-time select between two points, pick_b nonzero -> b (ref: point_448.h:558-563,
 * src/goldilocks.c:879-886), secure erase (ref: point_448.h:734-745, src/goldilocks.c:1332-1342);
 * goldilocks_448_point_copy is the inline above. */
GOLDILOCKS_AMD_API void goldilocks_448_point_cond_sel(goldilocks_448_point_p out,
        const goldilocks_448_point_p a, const goldilocks_448_point_p b, goldilocks_word_t pick_b);
GOLDILOCKS_AMD_API void goldilocks_448_point_destroy(goldilocks_448_point_p point);
GOLDILOCKS_AMD_API void goldilocks_448_precomputed_destroy(goldilocks_448_precomputed_s *pre);

/* RFC 8032 Ed448 verification.  ref: ed448.h:157-165, src/eddsa.c:253-306 */
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_ed448_verify(
        const uint8_t signature[GOLDILOCKS_EDDSA_448_SIGNATURE_BYTES],
        const uint8_t pubkey[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES],
        const uint8_t *message, size_t message_len, uint8_t prehashed,
        const uint8_t *context, uint8_t context_len);

/* --- "next" rows of SURVEY.md section 8(f): key derivation and signing (fixed-base comb + SHAKE256) --- */

/* pubkey = RFC 8032 public key of privkey.  ref: ed448.h:73-76, src/eddsa.c:131-147 */
GOLDILOCKS_AMD_API void goldilocks_ed448_derive_public_key(
        uint8_t pubkey[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES],
        const uint8_t privkey[GOLDILOCKS_EDDSA_448_PRIVATE_BYTES]);
/* RFC 8032 Ed448 / Ed448ph signature (prehashed != 0: message is the 64-byte prehash).
 * ref: ed448.h:95-104, src/eddsa.c:149-230 */
GOLDILOCKS_AMD_API void goldilocks_ed448_sign(
        uint8_t signature[GOLDILOCKS_EDDSA_448_SIGNATURE_BYTES],
        const uint8_t privkey[GOLDILOCKS_EDDSA_448_PRIVATE_BYTES],
        const uint8_t pubkey[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES],
        const uint8_t *message, size_t message_len, uint8_t prehashed,
        const uint8_t *context, uint8_t context_len);

/* --- "next" row f4: one base / two scalars, and Elligator 2 hash-to-curve --- */
/* a1 = scalar1*base1, a2 = scalar2*base1.  ref: point_448.h:517-523, src/goldilocks.c:543-642 */
GOLDILOCKS_AMD_API void goldilocks_448_point_dual_scalarmul(goldilocks_448_point_p a1, goldilocks_448_point_p a2,
        const goldilocks_448_point_p base1, const goldilocks_448_scalar_p scalar1,
        const goldilocks_448_scalar_p scalar2);
/* Elligator 2 map of one 56-byte string / sum of the maps of two.  ref: point_448.h:647-677, src/elligator.c:32-94 */
GOLDILOCKS_AMD_API void goldilocks_448_point_from_hash_nonuniform(goldilocks_448_point_p pt,
        const unsigned char hashed_data[GOLDILOCKS_448_SER_BYTES]);
GOLDILOCKS_AMD_API void goldilocks_448_point_from_hash_uniform(goldilocks_448_point_p pt,
        const unsigned char hashed_data[2 * GOLDILOCKS_448_SER_BYTES]);

/* --- "next" row f3: X448 (RFC 7748) --- */
/* shared = X448(scalar, base); FAILURE iff the result is all zero.  ref: point_448.h:398-402, src/goldilocks.c:1006-1076 */
GOLDILOCKS_AMD_API goldilocks_error_t goldilocks_x448(uint8_t shared[GOLDILOCKS_X448_PUBLIC_BYTES],
        const uint8_t base[GOLDILOCKS_X448_PUBLIC_BYTES], const uint8_t scalar[GOLDILOCKS_X448_PRIVATE_BYTES]);
/* out = X448(scalar, 5) computed on the fixed-base comb.  ref: point_448.h:445-448, src/goldilocks.c:1115-1141 */
GOLDILOCKS_AMD_API void goldilocks_x448_derive_public_key(uint8_t out[GOLDILOCKS_X448_PUBLIC_BYTES],
        const uint8_t scalar[GOLDILOCKS_X448_PRIVATE_BYTES]);
GOLDILOCKS_AMD_API extern const uint8_t goldilocks_x448_base_point[GOLDILOCKS_X448_PUBLIC_BYTES]; /* ref: point_448.h:413 */
/* out = (y / x)^2 of the internal point: the RFC 7748 encoding of GOLDILOCKS_X448_ENCODE_RATIO times the point (the base
 * point gives that multiple of the X448 base point); 1/0 = 0.  ref: point_448.h:404-427, src/goldilocks.c:1104-1115 */
GOLDILOCKS_AMD_API void goldilocks_448_point_mul_by_ratio_and_encode_like_x448(uint8_t out[GOLDILOCKS_X448_PUBLIC_BYTES],
        const goldilocks_448_point_p p);
/* The secret scalar of an Ed448 private key -- clamp(SHAKE256(privkey)[0:57]) mod q, divided by the encode ratio -- and its
 * X448 private key, SHAKE256(ed)[0:56].  ref: ed448.h:52-64, 250-263, src/eddsa.c:83-128 */
GOLDILOCKS_AMD_API void goldilocks_ed448_derive_secret_scalar(goldilocks_448_scalar_p secret,
        const uint8_t privkey[GOLDILOCKS_EDDSA_448_PRIVATE_BYTES]);
GOLDILOCKS_AMD_API void goldilocks_ed448_convert_private_key_to_x448(uint8_t x[GOLDILOCKS_X448_PRIVATE_BYTES],
        const uint8_t ed[GOLDILOCKS_EDDSA_448_PRIVATE_BYTES]);
/* x = y^2 (1 - d y^2) / (1 - y^2) of an Ed448 public key's y (its sign byte is not read).
 * ref: ed448.h:234-248, src/goldilocks.c:1079-1102 */
GOLDILOCKS_AMD_API void goldilocks_ed448_convert_public_key_to_x448(uint8_t x[GOLDILOCKS_X448_PUBLIC_BYTES],
        const uint8_t ed[GOLDILOCKS_EDDSA_448_PUBLIC_BYTES]);

/* ------------------------------------------------------------------ (2) host-array batches
 * All return 0 on success, nonzero on a runtime (HIP) error -- see goldilocks_amd_last_error().
 * Arrays are dense AoS of the reference structs; outputs may alias inputs of the same type. */

GOLDILOCKS_AMD_API int goldilocks_448_point_scalarmul_batch(goldilocks_448_point_s *scaled,
        const goldilocks_448_point_s *base, const goldilocks_448_scalar_s *scalar, size_t n);
GOLDILOCKS_AMD_API int goldilocks_448_precomputed_scalarmul_batch(goldilocks_448_point_s *scaled,
        const goldilocks_448_precomputed_s *table, const goldilocks_448_scalar_s *scalar, size_t n);
/* The same with the table access (GOLDILOCKS_AMD_CALL_TABLES_*, below) and the devices of THIS call: the batch
 * is cut into contiguous slices over devices[0..device_count) as goldilocks_amd_use_devices describes;
 * device_count = 0: the process-wide list (by default the calling thread's current device), devices == NULL
 * with device_count > 0: devices 0..device_count-1. */
GOLDILOCKS_AMD_API int goldilocks_448_point_scalarmul_batch_ex(goldilocks_448_point_s *scaled,
        const goldilocks_448_point_s *base, const goldilocks_448_scalar_s *scalar, size_t n, uint32_t flags,
        const int *devices, int device_count);
GOLDILOCKS_AMD_API int goldilocks_448_precomputed_scalarmul_batch_ex(goldilocks_448_point_s *scaled,
        const goldilocks_448_precomputed_s *table, const goldilocks_448_scalar_s *scalar, size_t n, uint32_t flags,
        const int *devices, int device_count);
GOLDILOCKS_AMD_API int goldilocks_448_point_double_scalarmul_batch(goldilocks_448_point_s *combo,
        const goldilocks_448_point_s *base1, const goldilocks_448_scalar_s *scalar1,
        const goldilocks_448_point_s *base2, const goldilocks_448_scalar_s *scalar2, size_t n);
GOLDILOCKS_AMD_API int goldilocks_448_point_encode_batch(uint8_t *ser /* n*56 */,
        const goldilocks_448_point_s *pt, size_t n);
GOLDILOCKS_AMD_API int goldilocks_448_point_decode_batch(goldilocks_448_point_s *pt,
        goldilocks_error_t *status, const uint8_t *ser /* n*56 */, goldilocks_bool_t allow_identity, size_t n);
GOLDILOCKS_AMD_API int goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa_batch(uint8_t *enc /* n*57 */,
        const goldilocks_448_point_s *pt, size_t n);
GOLDILOCKS_AMD_API int goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio_batch(goldilocks_448_point_s *pt,
        goldilocks_error_t *status, const uint8_t *enc /* n*57 */, size_t n);
/* status[i] = verify(sig[i], pk[i], msg[i], len[i], prehashed, ctx, ctxlen) */
GOLDILOCKS_AMD_API int goldilocks_ed448_verify_batch(goldilocks_error_t *status,
        const uint8_t *sig /* n*114 */, const uint8_t *pk /* n*57 */,
        const uint8_t *const *message, const size_t *message_len, uint8_t prehashed,
        const uint8_t *context, uint8_t context_len, size_t n);

/* ... over the devices of this call (verification is public data: no table-access choice) */
GOLDILOCKS_AMD_API int goldilocks_ed448_verify_batch_ex(goldilocks_error_t *status,
        const uint8_t *sig /* n*114 */, const uint8_t *pk /* n*57 */,
        const uint8_t *const *message, const size_t *message_len, uint8_t prehashed,
        const uint8_t *context, uint8_t context_len, size_t n, const int *devices, int device_count);

GOLDILOCKS_AMD_API int goldilocks_ed448_derive_public_key_batch(uint8_t *pubkey /* n*57 */,
        const uint8_t *privkey /* n*57 */, size_t n);
GOLDILOCKS_AMD_API int goldilocks_ed448_sign_batch(uint8_t *signature /* n*114 */,
        const uint8_t *privkey /* n*57 */, const uint8_t *pubkey /* n*57 */,
        const uint8_t *const *message, const size_t *message_len, uint8_t prehashed,
        const uint8_t *context, uint8_t context_len, size_t n);
/* status[i] as goldilocks_448_direct_scalarmul(scaled[i], base[i], scalar[i], ...); with short_circuit a
 * failing lane leaves scaled[i] untouched.  ref: point_448.h:378-384 */
GOLDILOCKS_AMD_API int goldilocks_448_direct_scalarmul_batch(uint8_t *scaled /* n*56 */,
        goldilocks_error_t *status, const uint8_t *base /* n*56 */, const goldilocks_448_scalar_s *scalar,
        goldilocks_bool_t allow_identity, goldilocks_bool_t short_circuit, size_t n);

GOLDILOCKS_AMD_API int goldilocks_448_point_dual_scalarmul_batch(goldilocks_448_point_s *a1,
        goldilocks_448_point_s *a2, const goldilocks_448_point_s *base, const goldilocks_448_scalar_s *scalar1,
        const goldilocks_448_scalar_s *scalar2, size_t n);
/* uniform == 0: n*56 bytes in; otherwise n*112 */
GOLDILOCKS_AMD_API int goldilocks_448_point_from_hash_batch(goldilocks_448_point_s *pt,
        const uint8_t *hashed_data, int uniform, size_t n);
/* out[i] = point_mul_by_ratio_and_encode_like_x448(pt[i]); x[i] = ed448_convert_public_key_to_x448(ed[i]) */
GOLDILOCKS_AMD_API int goldilocks_448_point_mul_by_ratio_and_encode_like_x448_batch(uint8_t *out /* n*56 */,
        const goldilocks_448_point_s *pt, size_t n);
/* the scalar operations of goldilocks_amd_scalar_op_dev on host arrays (status: goldilocks_error_t[n] for ops 4 and 5, or NULL) */
GOLDILOCKS_AMD_API int goldilocks_amd_scalar_op_batch(goldilocks_448_scalar_s *out, goldilocks_error_t *status, const void *a,
        const goldilocks_448_scalar_s *b, int op, size_t len, size_t n);
GOLDILOCKS_AMD_API int goldilocks_ed448_convert_public_key_to_x448_batch(uint8_t *x /* n*56 */, const uint8_t *ed /* n*57 */,
        size_t n);
GOLDILOCKS_AMD_API int goldilocks_ed448_derive_secret_scalar_batch(goldilocks_448_scalar_s *secret,
        const uint8_t *privkey /* n*57 */, size_t n);
GOLDILOCKS_AMD_API int goldilocks_ed448_convert_private_key_to_x448_batch(uint8_t *x /* n*56 */, const uint8_t *ed /* n*57 */,
        size_t n);
/* status[i] = goldilocks_x448(shared[i], base[i], scalar[i]); base == NULL: the base point (derive_public_key) */
GOLDILOCKS_AMD_API int goldilocks_x448_batch(uint8_t *shared /* n*56 */, goldilocks_error_t *status,
        const uint8_t *base /* n*56 or NULL */, const uint8_t *scalar /* n*56 */, size_t n);

/* ------------------------------------------------------------------ (3) device-array API
 * Pointers are device pointers (hipMalloc / torch.Tensor.data_ptr()); `stream` is a
 * hipStream_t (NULL = default stream).  Calls are asynchronous on that stream and use a
 * per-device workspace owned by the library; a call on a different stream than the previous
 * workspace user first waits (on the device, hipStreamWaitEvent) for that one to finish, so calls
 * from several streams are safe and simply serialize where they share the workspace. */

/* Bind this process to a device and build the device-resident tables.  Optional (every
 * entry point initialises lazily on the current HIP device).  Returns 0 on success. */
GOLDILOCKS_AMD_API int goldilocks_amd_init(int device);
GOLDILOCKS_AMD_API void goldilocks_amd_shutdown(void);
GOLDILOCKS_AMD_API const char *goldilocks_amd_last_error(void);
/* the toolchain that compiled this library ("clang <version>; HIP <version>; <flags>"): a static string */
GOLDILOCKS_AMD_API const char *goldilocks_amd_build_info(void);
/* Multi-GPU for the host-array batches (independent operations, contiguous slice
 * [g*n/G, (g+1)*n/G) per GPU, one host thread per GPU, no cross-device traffic): after this call
 * goldilocks_448_point_scalarmul_batch, goldilocks_448_precomputed_scalarmul_batch and
 * goldilocks_ed448_verify_batch split every batch over the listed HIP devices.  count = 0 restores
 * the default (the calling thread's current device only); devices == NULL with count > 0 means
 * devices 0..count-1.  A device may be listed more than once (its shards then run one after the
 * other).  Returns 0 on success. */
GOLDILOCKS_AMD_API int goldilocks_amd_use_devices(const int *devices, int count);
/* Table-access policy for every multiplication whose scalar may be SECRET.  The reference reads its
 * window and comb tables with constant_time_lookup (src/include/constant_time.h:134-183; contract in its
 * README.md:92-97: no secret-dependent branches or memory addresses) in point_scalarmul,
 * point_double_scalarmul, point_dual_scalarmul, direct_scalarmul, precomputed_scalarmul and through
 * them derive_public_key, sign and x448_derive_public_key.  This library keeps that contract BY DEFAULT:
 *
 *   GOLDILOCKS_AMD_TABLES_INDEX_INDEPENDENT (default)
 *     - base-point multiplications (derive, sign, X448 key generation, precomputed_scalarmul on the
 *       built-in table): a signed comb of the base point staged in LDS (4 combs x 7 teeth x spacing 16,
 *       the structure of the reference's 5x5x18 with fewer additions), every lookup a
 *       wavefront-shuffle gather whose addresses and timing do not depend on the digit;
 *     - every variable-base multiplication (goldilocks_448_point_scalarmul, direct_scalarmul,
 *       point_double_scalarmul, point_dual_scalarmul): NO table -- a Montgomery ladder on the Montgomery
 *       model of the curve with selects only, then the other coordinate is recovered
 *       (csrc/montgomery.hpp; double_ and dual_scalarmul run two such ladders).  RESULT CONTRACT: this
 *       mode computes (s mod q) * P, the reference's recoding (s + m q) * P for an m of its own
 *       (src/goldilocks.c:420-438); on the points the API produces (the subgroup 2E) the two agree up to the
 *       2-torsion point (0, -1), i.e. results are defined up to goldilocks_448_point_eq and the encodings
 *       (which is what the reference's own backends guarantee about raw limbs too), not as raw limbs; the
 *       identity and (0, -1) as base or result come back as the identity's limbs.
 *     - batches small enough for one operation per wavefront read all 16 entries of an LDS window table for
 *       every digit and keep one by select.
 *     tools/isa_audit.py checks the compiled gfx950 code of every kernel of this mode: no branch
 *     condition and no memory address depends on the scalar (tests/test_isa_audit.py).
 *   GOLDILOCKS_AMD_TABLES_FAST (opt-in, for PUBLIC scalars only)
 *     - each lookup reads only the digit's entry (the address depends on the digit): the base point's
 *       window table in global memory (goldilocks_amd_set_base_table_bits); 5-bit windows per lane for
 *       goldilocks_448_point_double_scalarmul (two tables on ONE doubling chain instead of two ladders: 42 against
 *       62 ms per 2^20).  goldilocks_448_point_scalarmul, _direct_scalarmul and _point_dual_scalarmul run the
 *       table-free ladder(s) in THIS mode too since round 6: the ladder is the faster way to multiply one variable base
 *       (32.8 against 32.4 M/s; dual: 61.7 against 62.7 ms) and holds no 544-MiB table workspace.
 *
 * Not affected: verification and base_double_scalarmul_non_secret (public by contract: always the
 * fast tables), goldilocks_x448 with a peer's point (a Montgomery ladder with selects, no table),
 * caller-supplied precomputed_s tables (always an LDS comb with the same gather: the reference's 5x5x18,
 * re-combed to 4x7x16 per call for batches of 2^18 operations or more).
 *
 * goldilocks_amd_set_table_access sets the PROCESS-WIDE DEFAULT only (returns 0, or nonzero for an unknown
 * mode).  The reference has no mutable global state on these paths (every function is reentrant), so a
 * caller with both public and secret scalars in one process should leave the default alone and say what it
 * wants PER CALL: every entry point whose table access depends on the mode has an *_ex twin taking `flags`,
 *     GOLDILOCKS_AMD_CALL_TABLES_DEFAULT             the process-wide default (what the plain name does)
 *     GOLDILOCKS_AMD_CALL_TABLES_FAST                this call's scalars are public
 *     GOLDILOCKS_AMD_CALL_TABLES_INDEX_INDEPENDENT   this call's scalars may be secret
 * so that one thread's opt-in for public data can never downgrade another thread's signing. */
#define GOLDILOCKS_AMD_TABLES_FAST 0
#define GOLDILOCKS_AMD_TABLES_INDEX_INDEPENDENT 1
GOLDILOCKS_AMD_API int goldilocks_amd_set_table_access(int mode);
GOLDILOCKS_AMD_API int goldilocks_amd_get_table_access(void);   /* the process-wide default in force */
#define GOLDILOCKS_AMD_CALL_TABLES_DEFAULT 0u
#define GOLDILOCKS_AMD_CALL_TABLES_FAST 1u
#define GOLDILOCKS_AMD_CALL_TABLES_INDEX_INDEPENDENT 2u
#define GOLDILOCKS_AMD_CALL_TABLES_MASK 3u
/* Test hook: how many mode-dependent calls of the CALLING THREAD ran with digit-addressed tables (counts[0]) and
 * index-independently (counts[1]) so far -- what a call's flags and the default resolved to when it launched. */
GOLDILOCKS_AMD_API void goldilocks_amd_thread_mode_counts(uint64_t counts[2]);
/* Small batches.  One lane's ladder takes 2.1-2.8 ms however few operations a call has, so batches of
 * up to `n` variable-base, double-base or dual multiplications -- 3n/4 fixed-base multiplications or X448
 * shared secrets, n/2 verifications, wire-format multiplications, key derivations, X448 key generations
 * or comb tables (precompute), n/4 signatures: the measured crossovers, tools/probes/crossover_probe.py -- and
 * up to 1024 encodings, decodings or hash-to-curve maps, the single-operation drop-in names included,
 * run ONE OPERATION PER WAVEFRONT instead: the 64 lanes share the operation (a field element spread
 * over the 16 lanes of a row, four field elements per register), 0.35 ms per multiplication call,
 * 0.57 ms per verification call, 0.43 ms per signature, 0.16 ms per encoding or decoding.
 * Index-independent table access in either table mode.  The default is the measured crossover
 * (profiles/r02/wave_probe.txt); 0 disables the path. */
#define GOLDILOCKS_AMD_WAVE_BATCH_DEFAULT 8192
#define GOLDILOCKS_AMD_WAVE_VERIFY_DEFAULT (GOLDILOCKS_AMD_WAVE_BATCH_DEFAULT / 2)
GOLDILOCKS_AMD_API void goldilocks_amd_set_wave_batch_max(size_t n);
GOLDILOCKS_AMD_API size_t goldilocks_amd_get_wave_batch_max(void);
/* Verification batches usually hold many signatures of few keys.  For batches of at least `min_batch` signatures the
 * device entry point therefore decodes every DISTINCT public key of the batch once and builds its window table once, in
 * a pool of up to `keys` tables (4 KiB each) that the lanes share; a batch with more distinct keys than that, or in
 * which more than half of the signatures bring a key of their own, uses no pool (every lane decodes its key and builds
 * its table itself, as small batches do).  Verdicts do not change.  keys = 0 turns the pool off.  Process-wide. */
#define GOLDILOCKS_AMD_KEY_POOL_DEFAULT (1u << 18)
#define GOLDILOCKS_AMD_KEY_POOL_MIN_BATCH_DEFAULT (1u << 16)
GOLDILOCKS_AMD_API void goldilocks_amd_set_verify_key_pool(size_t keys, size_t min_batch);
/* Keys that sign MANY signatures of a batch get more than a shared window table: a fixed-base comb of their own (4 x 7 x
 * 16, 48 KiB per key, built on the device per call), with which a verification is src/eddsa.c's equation without a
 * ladder and without R's decoding -- 0.3 of the arithmetic.  Used when the batch averages at least
 * `min_signatures_per_key` signatures per distinct key (twice that for batches below 2^18 signatures, four times below 2^17,
 * where the fixed latency of building the combs weighs more: 8 / 16 / 32 by default, the measured break-evens of
 * tools/probes/key_pool_probe.py) and has at most `keys` distinct keys, in batches of more than 2^12
 * signatures (whatever the pool's min_batch); otherwise the pool's rules apply.  Verdicts do not change.  keys = 0 turns the
 * combs off; 2^17 is the most and the default.  `keys` is a CEILING: a batch can use at most n / min_signatures_per_key
 * combs, so that is what a call reserves workspace for -- 71 KiB of device memory per such key, never more than
 * goldilocks_amd_set_verify_key_combs_bytes (4 GiB by default: 59 000 keys) nor than a quarter of the device's free memory,
 * kept until goldilocks_amd_shutdown -- whatever the batch's keys then turn out to be (the device decides; the call does
 * not wait for it).  A batch with more distinct keys than fit is served by the pool's rules.  Turning the pool off
 * (goldilocks_amd_set_verify_key_pool(0, ..)) turns the combs off with it.  Process-wide. */
#define GOLDILOCKS_AMD_KEY_COMBS_DEFAULT (1u << 17)
#define GOLDILOCKS_AMD_KEY_COMBS_MIN_PER_KEY_DEFAULT 8
GOLDILOCKS_AMD_API void goldilocks_amd_set_verify_key_combs(size_t keys, size_t min_signatures_per_key);
/* the ceiling on the workspace a call reserves for per-key combs (2^17 keys, the most the library serves, take 9 GiB) */
#define GOLDILOCKS_AMD_KEY_COMBS_BYTES_DEFAULT ((size_t)4 << 30)
GOLDILOCKS_AMD_API void goldilocks_amd_set_verify_key_combs_bytes(size_t bytes);
/* ... and keys that sign at least this many signatures of the batch on average get the wider comb (4 x 8 x 14, 96 KiB:
 * twice the entries to build, 9 % less to walk per signature).  0: never.  Process-wide. */
#define GOLDILOCKS_AMD_KEY_COMBS_WIDE_MIN_PER_KEY_DEFAULT 256
GOLDILOCKS_AMD_API void goldilocks_amd_set_verify_key_combs_wide(size_t min_signatures_per_key);
/* ... and from this many on the widest (5 x 9 x 10, 240 KiB: two and a half times the entries again, another 10 % less to
 * walk per signature: 9 doublings + 49 additions).  0: never.  Process-wide. */
#define GOLDILOCKS_AMD_KEY_COMBS_XWIDE_MIN_PER_KEY_DEFAULT 1024
GOLDILOCKS_AMD_API void goldilocks_amd_set_verify_key_combs_xwide(size_t min_signatures_per_key);
/* The base point's own window table: T_i[k] = (2k+1) * 2^(w i) * B for the signed w-bit digits of a scalar, one mixed
 * addition per digit and no doubling.  It serves what multiplies the base point by PUBLIC data -- S*B of every
 * verification, goldilocks_448_base_double_scalarmul_non_secret -- and, with GOLDILOCKS_AMD_TABLES_FAST, key derivation,
 * signing, X448 key generation and the base point's precomputed_scalarmul.  A device builds it at the first call that
 * needs it (tens of milliseconds) and keeps it until goldilocks_amd_shutdown.  Wider digits trade device memory for
 * additions: 16 bits = 27 additions from 168 MiB (Infinity-Cache resident), 18 = 24 from 600 MiB, 20 = 22 from 2.2 GiB,
 * 22 = 20 from 7.9 GiB, 24 = 18 from 28.5 GiB of HBM (2^20 verifications of distinct signatures on 2^10 keys, one box,
 * round 5: 7.63 / 7.54 / 7.44 / 7.32 / 7.24 ms at 16 / 18 / 20 / 22 / 24 bits; base-point multiplications with
 * digit-addressed tables: 507 / 544 / 632 M/s at 16 / 20 / 24; profiles/r05/bench_verify_widths.txt).
 * bits = 0 (the default) is GOLDILOCKS_AMD_BASE_TABLE_BITS_DEFAULT = 20: 2.2 GiB whatever else the device holds -- the
 * wider tables buy 2 - 4 % for up to 26 GiB more and are the caller's decision.  GOLDILOCKS_AMD_BASE_TABLE_BITS_AUTO asks
 * for the widest whose table takes at most an eighth of the device memory free at that first call, never below 16 (24 bits
 * on an otherwise empty MI355X).  Any even width from 8 to 24 may be asked for.  The environment's
 * GOLDILOCKS_AMD_BASE_TABLE_BITS (a width, or "auto") sets the process's initial request.  Process-wide; a device whose
 * table has another width rebuilds it at its next such call (which then waits for the device, also on the
 * stream-asynchronous _dev entry points: hipDeviceSynchronize + the build).  Results do not depend on the width.
 * Returns 0, or 1 for a width that does not exist.
 *
 * DEVICE MEMORY THE LIBRARY HOLDS at these defaults, until goldilocks_amd_shutdown or goldilocks_amd_release_memory:
 * the base point's table 2.2 GiB + the workspace, which grows with the largest batch seen: at most 7.5 GiB for 2^20
 * verifications (measured 7.0: the per-key combs' ceiling of 4 GiB above, 2 GiB of pooled per-key tables, 1 GiB of
 * per-lane tables and bookkeeping), 128 MiB for 2^20 variable-base multiplications -- 10 GiB in all, asserted by
 * tests/test_gpu_base_table.py -- + the host-array entry points' staging (1.2 x the batch's bytes). */
#define GOLDILOCKS_AMD_BASE_TABLE_BITS_DEFAULT 20
#define GOLDILOCKS_AMD_BASE_TABLE_BITS_AUTO 1
GOLDILOCKS_AMD_API int goldilocks_amd_set_base_table_bits(int bits);
/* the width of the table the calling thread's device holds (0: none built yet) */
GOLDILOCKS_AMD_API int goldilocks_amd_get_base_table_bits(void);
/* Test hook: how the last large verification batch on the calling thread's device served its keys --
 * counts[0] distinct keys seen, counts[1] keys with a pooled window table, counts[2] keys with a comb (at most one of the
 * two is non-zero; all zero if the batch was too small for either), counts[3] the combs' teeth (7, 8 or 9; 0 without
 * combs).  Waits for the device. */
GOLDILOCKS_AMD_API int goldilocks_amd_last_verify_key_counts(uint32_t counts[4]);
/* Test hook: entries [first, first + count) of the base point's window table on the calling thread's device (built at the
 * width in force if it is not there yet) as canonical bytes -- a, b, cn of each affine niels, 3 x 56 bytes -- into host
 * memory: the every-entry check of the table's build (tests/test_gpu_every_lane.py).  Waits for the device. */
GOLDILOCKS_AMD_API int goldilocks_amd_base_table_export(uint8_t *dst /* count*168 */, uint64_t first, size_t count);
/* "gfx950", number of CUs, device memory the library currently holds on the calling thread's device besides its small
 * tables: workspace + staging + the base point's window table */
GOLDILOCKS_AMD_API int goldilocks_amd_device_info(char *arch, size_t arch_len, int *compute_units,
        size_t *workspace_bytes);
/* Gives device memory of the calling thread's device back without ending the context: the library keeps its workspace
 * (sized by the largest batch so far: at most 7.5 GiB after 2^20 verifications), the staging buffers of the host-array entry points
 * and the base point's window table (2.2 GiB by default) from their first use until goldilocks_amd_shutdown, because
 * allocating them costs milliseconds per call.  A service that is done with a burst can release any of them; the next call
 * that needs one allocates (and, for the table, builds: 0.16 s) it again.  Waits for the device.  Returns 0 on success. */
#define GOLDILOCKS_AMD_RELEASE_WORKSPACE 1u
#define GOLDILOCKS_AMD_RELEASE_STAGING 2u
#define GOLDILOCKS_AMD_RELEASE_BASE_TABLE 4u
#define GOLDILOCKS_AMD_RELEASE_ALL 7u
GOLDILOCKS_AMD_API int goldilocks_amd_release_memory(uint32_t what);

GOLDILOCKS_AMD_API int goldilocks_amd_point_scalarmul_dev(void *scaled /* point_s[n] */,
        const void *base /* point_s[n] */, const void *scalar /* scalar_s[n] */, size_t n, void *stream);
/* table == NULL selects the built-in base-point comb; otherwise a DEVICE copy of a precomputed_s */
GOLDILOCKS_AMD_API int goldilocks_amd_precomputed_scalarmul_dev(void *scaled, const void *table,
        const void *scalar, size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_double_scalarmul_dev(void *combo, const void *base1,
        const void *scalar1, const void *base2, const void *scalar2, size_t n, void *stream);
/* base1 == point_base for every lane (verify's shape) */
GOLDILOCKS_AMD_API int goldilocks_amd_base_double_scalarmul_dev(void *combo, const void *scalar1,
        const void *base2, const void *scalar2, size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_encode_dev(void *ser /* n*56 B */, const void *pt, size_t n,
        void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_decode_dev(void *pt, void *status /* int32[n] */,
        const void *ser, int allow_identity, size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_encode_eddsa_dev(void *enc /* n*57 B */, const void *pt, size_t n,
        void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_decode_eddsa_dev(void *pt, void *status /* int32[n] */,
        const void *enc, size_t n, void *stream);
/* op: 0 add, 1 sub, 2 double, 3 negate (b ignored), 4 debugging_torque (b ignored), 5 debugging_pscale (b: 56 bytes per
 * point, the factors); out/a/b: point_s[n] */
GOLDILOCKS_AMD_API int goldilocks_amd_point_op_dev(void *out, const void *a, const void *b, int op, size_t n,
        void *stream);
/* Scalars mod q, n operations.  op: 0 add, 1 sub, 2 mul (out, a, b: scalar_s[n]), 3 halve, 4 invert (status: int32[n],
 * success iff a != 0), 5 decode (a: 56 bytes per operation; status: success iff below q), 6 decode_long (a: len bytes per
 * operation).  status may be NULL for the other operations.  ref: src/scalar.c */
GOLDILOCKS_AMD_API int goldilocks_amd_scalar_op_dev(void *out, void *status, const void *a, const void *b, int op,
        size_t len, size_t n, void *stream);
/* op: 0 eq(a,b), 1 valid(a); status: int32[n] (-1 / 0) */
GOLDILOCKS_AMD_API int goldilocks_amd_point_pred_dev(void *status, const void *a, const void *b, int op,
        size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_precompute_dev(void *table /* precomputed_s[n] */, const void *base,
        size_t n, void *stream);
/* Messages: if msg_offsets != NULL, message i is msgs[msg_offsets[i] .. msg_offsets[i+1])
 * (uint64 offsets, n+1 entries, device memory); otherwise every message is msg_len bytes at
 * msgs + i*msg_len.  ctx: device pointer to ctx_len bytes (may be NULL when ctx_len == 0).
 * status: int32[n], -1 = GOLDILOCKS_SUCCESS, 0 = GOLDILOCKS_FAILURE.
 * Every message must be shorter than GOLDILOCKS_AMD_MAX_MESSAGE_BYTES: the host-array calls and the
 * fixed-length form return an error for longer ones; a longer message behind device-resident offsets
 * makes that lane FAIL verification (and signing writes an all-zero signature for it). */
#define GOLDILOCKS_AMD_MAX_MESSAGE_BYTES 0x7fffff00ull
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_verify_dev(void *status, const void *sig, const void *pk,
        const void *msgs, const void *msg_offsets, size_t msg_len, uint8_t prehashed,
        const void *ctx, uint8_t ctx_len, size_t n, void *stream);

GOLDILOCKS_AMD_API int goldilocks_amd_ed448_derive_public_key_dev(void *pubkey /* n*57 */,
        const void *privkey /* n*57 */, size_t n, void *stream);
/* message layout as for goldilocks_amd_ed448_verify_dev */
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_sign_dev(void *signature /* n*114 */, const void *privkey,
        const void *pubkey, const void *msgs, const void *msg_offsets, size_t msg_len, uint8_t prehashed,
        const void *ctx, uint8_t ctx_len, size_t n, void *stream);
/* status: int32[n] */
GOLDILOCKS_AMD_API int goldilocks_amd_direct_scalarmul_dev(void *scaled /* n*56 */, void *status,
        const void *base /* n*56 */, const void *scalar, int allow_identity, int short_circuit, size_t n,
        void *stream);

GOLDILOCKS_AMD_API int goldilocks_amd_point_dual_scalarmul_dev(void *a1, void *a2, const void *base,
        const void *scalar1, const void *scalar2, size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_point_from_hash_dev(void *pt, const void *hashed_data, int uniform,
        size_t n, void *stream);
/* the two conversions above over device arrays */
GOLDILOCKS_AMD_API int goldilocks_amd_point_encode_like_x448_dev(void *out /* n*56 */, const void *pt /* n*256 */, size_t n,
        void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_convert_public_key_to_x448_dev(void *x /* n*56 */, const void *ed /* n*57 */,
        size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_derive_secret_scalar_dev(void *secret /* n*56 */, const void *privkey /* n*57 */,
        size_t n, void *stream);
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_convert_private_key_to_x448_dev(void *x /* n*56 */, const void *ed /* n*57 */,
        size_t n, void *stream);
/* base == NULL: x448_derive_public_key for every lane (status all success); status: int32[n] or NULL */
GOLDILOCKS_AMD_API int goldilocks_amd_x448_dev(void *shared /* n*56 */, void *status, const void *base,
        const void *scalar /* n*56 */, size_t n, void *stream);

/* The device-array entry points whose table access depends on the mode, with the mode of THIS call
 * (flags = GOLDILOCKS_AMD_CALL_TABLES_*; unknown bits are an error).  Everything else as the plain name. */
GOLDILOCKS_AMD_API int goldilocks_amd_point_scalarmul_dev_ex(void *scaled, const void *base, const void *scalar,
        size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_precomputed_scalarmul_dev_ex(void *scaled, const void *table,
        const void *scalar, size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_point_double_scalarmul_dev_ex(void *combo, const void *base1,
        const void *scalar1, const void *base2, const void *scalar2, size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_derive_public_key_dev_ex(void *pubkey, const void *privkey, size_t n,
        void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_ed448_sign_dev_ex(void *signature, const void *privkey, const void *pubkey,
        const void *msgs, const void *msg_offsets, size_t msg_len, uint8_t prehashed, const void *ctx,
        uint8_t ctx_len, size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_direct_scalarmul_dev_ex(void *scaled, void *status, const void *base,
        const void *scalar, int allow_identity, int short_circuit, size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_point_dual_scalarmul_dev_ex(void *a1, void *a2, const void *base,
        const void *scalar1, const void *scalar2, size_t n, void *stream, uint32_t flags);
GOLDILOCKS_AMD_API int goldilocks_amd_x448_dev_ex(void *shared, void *status, const void *base, const void *scalar,
        size_t n, void *stream, uint32_t flags);

/* Field-level test hook (parity tests for SURVEY 8a rows a2-a7; ref: src/f_field.h:76-79,
 * src/arch_ref64/f_impl.h:10-38, src/f_generic.c:19-131).  a, b, out: gf_448_s[n] in the ABI limb form.
 * op & 0xff:  0 out = a*b   1 out = a^2   2 out = isr(a), status = mask   3 out = strong_reduce(a), raw limbs
 *             4 out = a*w, w = low 32 bits of b[i].limb[0] (gf_mulw_unsigned)
 *             5 out = gf_add(a, b)   6 out = gf_sub(a, b)   7 out = weak_reduce(a)   (inputs taken as
 *               given, unreduced limbs included; outputs raw limbs)
 *             8 status = gf_eq(a, b)   9 status = gf_lobit(a)
 *            10 out = the 56 serialized bytes (limbs 0..6)   11 a = 56 bytes, out = limbs, status = value < p
 *            12 out = (ma*a)*(mb*b), 13 out = (ma*a)^2 with the multiples formed limb-wise without
 *               reduction, ma = (op >> 8) & 0xff, mb = (op >> 16) & 0xff: operands at the limits of the
 *               device arithmetic's magnitude contract.
 *            14 out = (ka*a)*(kb*b), 15 out = (ka*a)^2 through the SIGNED, register-paired layer of the ladders
 *               (csrc/gf28s.hpp): a multiplicity is 1 .. 3 copies added pair-wise, or with bit 7 set the limb-wise
 *               negative of that many (a difference-like operand); 15 with (op >> 16) & 0xff != 0: the square of a sum
 *               of products (its unsigned-offset columns). */
GOLDILOCKS_AMD_API int goldilocks_amd_field_op_dev(void *out, void *status, const void *a, const void *b,
        int op, size_t n, void *stream);

/* The same hook for the row arithmetic of the one-operation-per-wave path (four elements per wavefront):
 * op 0 mul, 1 strong_reduce (raw limbs), 2 isr (+ mask), 3 eq (mask), 4 lobit (mask),
 * 5 deserialize (a = 56 bytes; out = limbs, status = value < p). */
GOLDILOCKS_AMD_API int goldilocks_amd_wave_field_op_dev(void *out, void *status, const void *a, const void *b,
        int op, size_t n, void *stream);

/* Test hook for verification's half-size scalars (csrc/lattice.hpp; no counterpart in the reference, which
 * walks the full challenge, src/eddsa.c:283-327): for each challenge h[i] < q (goldilocks_448_scalar_s)
 * rho[i] = 15 little-endian uint32 words, 0 <= rho < 2^223, and tau[i] = 8 words, two's complement,
 * 0 < |tau| < 2^223, with rho == tau * h (mod q): the first pair below 2^223 of the remainder sequence of (q, h). */
GOLDILOCKS_AMD_API int goldilocks_amd_half_size_pair_dev(void *rho /* n*15 uint32 */, void *tau /* n*8 uint32 */,
        const void *h, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GOLDILOCKS_AMD_H */
