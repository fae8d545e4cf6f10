/*
 * gold_oracle.h -- CPU ORACLE for the Ed448-Goldilocks hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is a from-scratch, plain-C restatement of
 * the algorithms of otrv4/libgoldilocks (arch_ref64 shape: 8 x 56-bit limbs,
 * unsigned __int128 accumulators) for the batched-scalarmul hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product library (libgoldilocks_amd) never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check every function here
 * against (1) the reference's own known-answer vectors (RFC 8032 Ed448 x11,
 * base_multiples k*B k<16 from test/elligator_vectors.inc.cxx, RFC 7748
 * independent constants), (2) golden fixtures generated in the build
 * container from the real reference compiled as oracle/_ref (generator:
 * tests/golden/gen_golden.py), and (3) when oracle/_ref is present, live
 * differential runs against the real reference.
 *
 * All types are layout-compatible with the reference's public ABI
 * (src/public_include/goldilocks/point_448.h:33-35, 66-70, 82-86).
 */
#ifndef GOLD_ORACLE_H
#define GOLD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct { uint64_t limb[8]; } __attribute__((aligned(32))) orc_gf;
typedef struct { orc_gf x, y, z, t; } orc_point;
typedef struct { uint64_t limb[7]; } orc_scalar;
typedef struct { orc_gf a, b, c; } orc_niels;
typedef struct { orc_niels n; orc_gf z; } orc_pniels;
typedef struct { orc_niels table[80]; } orc_precomputed;

#define ORC_SUCCESS (-1)
#define ORC_FAILURE (0)

/* ---- field GF(2^448 - 2^224 - 1) ---- */
ORC_API void orc_gf_mul(orc_gf *c, const orc_gf *a, const orc_gf *b);
ORC_API void orc_gf_sqr(orc_gf *c, const orc_gf *a);
ORC_API void orc_gf_mulw(orc_gf *c, const orc_gf *a, uint32_t w);
ORC_API void orc_gf_add(orc_gf *c, const orc_gf *a, const orc_gf *b);
ORC_API void orc_gf_sub(orc_gf *c, const orc_gf *a, const orc_gf *b);
ORC_API void orc_gf_strong_reduce(orc_gf *a);
ORC_API uint64_t orc_gf_isr(orc_gf *out, const orc_gf *x);   /* mask */
ORC_API void orc_gf_serialize(uint8_t out[56], const orc_gf *x);
ORC_API uint64_t orc_gf_deserialize(orc_gf *x, const uint8_t in[56], uint8_t hi_nmask);
ORC_API uint64_t orc_gf_eq(const orc_gf *a, const orc_gf *b);
ORC_API uint64_t orc_gf_lobit(const orc_gf *a);

/* ---- scalars mod q ---- */
ORC_API void orc_scalar_add(orc_scalar *o, const orc_scalar *a, const orc_scalar *b);
ORC_API void orc_scalar_sub(orc_scalar *o, const orc_scalar *a, const orc_scalar *b);
ORC_API void orc_scalar_mul(orc_scalar *o, const orc_scalar *a, const orc_scalar *b);
ORC_API void orc_scalar_halve(orc_scalar *o, const orc_scalar *a);
ORC_API int  orc_scalar_invert(orc_scalar *o, const orc_scalar *a);   /* scalar.c:107-166 */
ORC_API int  orc_scalar_decode(orc_scalar *o, const uint8_t in[56]);
ORC_API void orc_scalar_decode_long(orc_scalar *o, const uint8_t *in, size_t len);
ORC_API void orc_scalar_encode(uint8_t out[56], const orc_scalar *a);

/* ---- group ---- */
ORC_API const orc_point *orc_point_base(void);
ORC_API const orc_point *orc_point_identity(void);
ORC_API const orc_precomputed *orc_precomputed_base(void);
ORC_API const orc_niels *orc_wnaf_base(void);      /* 32 affine niels */

ORC_API void orc_point_add(orc_point *p, const orc_point *q, const orc_point *r);
ORC_API void orc_point_sub(orc_point *p, const orc_point *q, const orc_point *r);
ORC_API void orc_point_double(orc_point *p, const orc_point *q);
ORC_API void orc_point_negate(orc_point *p, const orc_point *q);
ORC_API void orc_point_debugging_torque(orc_point *q, const orc_point *p);                         /* goldilocks.c:675-683 */
ORC_API void orc_point_debugging_pscale(orc_point *q, const orc_point *p, const uint8_t factor[56]);  /* goldilocks.c:685-701 */
ORC_API int  orc_point_eq(const orc_point *p, const orc_point *q);       /* -1 / 0 */
ORC_API int  orc_point_valid(const orc_point *p);                       /* -1 / 0 */
ORC_API void orc_point_encode(uint8_t out[56], const orc_point *p);
ORC_API int  orc_point_decode(orc_point *p, const uint8_t in[56], int allow_identity);
ORC_API void orc_point_encode_like_eddsa(uint8_t out[57], const orc_point *p);
ORC_API int  orc_point_decode_like_eddsa(orc_point *p, const uint8_t in[57]);

ORC_API void orc_point_scalarmul(orc_point *out, const orc_point *base, const orc_scalar *s);
ORC_API void orc_precompute(orc_precomputed *tab, const orc_point *base);
ORC_API void orc_precompute_wnafs(orc_niels out[32], const orc_point *base);
ORC_API void orc_precomputed_scalarmul(orc_point *out, const orc_precomputed *tab, const orc_scalar *s);
ORC_API void orc_point_double_scalarmul(orc_point *out, const orc_point *b, const orc_scalar *sb,
                                        const orc_point *c, const orc_scalar *sc);
ORC_API void orc_base_double_scalarmul_non_secret(orc_point *out, const orc_scalar *s1,
                                        const orc_point *base2, const orc_scalar *s2);
ORC_API int  orc_direct_scalarmul(uint8_t out[56], const uint8_t base[56], const orc_scalar *s,
                                  int allow_identity, int short_circuit);

/* ---- Elligator 2 hash-to-curve ---- */
ORC_API void orc_point_from_hash_nonuniform(orc_point *p, const uint8_t ser[56]);
ORC_API void orc_point_from_hash_uniform(orc_point *p, const uint8_t ser[112]);

/* ---- X448 (RFC 7748) ---- */
ORC_API int  orc_x448(uint8_t out[56], const uint8_t base[56], const uint8_t scalar[56]);
ORC_API void orc_x448_derive_public_key(uint8_t out[56], const uint8_t scalar[56]);
ORC_API void orc_point_encode_like_x448(uint8_t out[56], const orc_point *p);
ORC_API void orc_ed448_convert_public_key_to_x448(uint8_t x[56], const uint8_t ed[57]);
ORC_API void orc_ed448_derive_secret_scalar(orc_scalar *s, const uint8_t sk[57]);
ORC_API void orc_ed448_convert_private_key_to_x448(uint8_t x[56], const uint8_t ed[57]);

/* ---- SHAKE256 / EdDSA ---- */
ORC_API void orc_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
ORC_API void orc_ed448_derive_public_key(uint8_t pk[57], const uint8_t sk[57]);
ORC_API void orc_ed448_sign(uint8_t sig[114], const uint8_t sk[57], const uint8_t pk[57],
                            const uint8_t *msg, size_t msglen, uint8_t prehashed,
                            const uint8_t *ctx, uint8_t ctxlen);
ORC_API int  orc_ed448_verify(const uint8_t sig[114], const uint8_t pk[57],
                              const uint8_t *msg, size_t msglen, uint8_t prehashed,
                              const uint8_t *ctx, uint8_t ctxlen);

/* ---- batch drivers (pthreads) used for cpu_baseline timing and bulk checks ---- */
ORC_API void orc_point_scalarmul_batch(orc_point *out, const orc_point *base,
                                       const orc_scalar *s, size_t n, int nthreads);
ORC_API void orc_precomputed_scalarmul_batch(orc_point *out, const orc_precomputed *tab,
                                       const orc_scalar *s, size_t n, int nthreads);
/* times an external (e.g. the real reference's) scalarmul function over a batch */
ORC_API void orc_extern_scalarmul_batch(void (*fn)(void *, const void *, const void *), orc_point *out,
                                        const orc_point *base, const orc_scalar *s, size_t n, int nthreads);
/* reference-style timing (test/bench_goldilocks.cxx:73-143): nsamples x ntests calls, one thread;
 * fn == NULL times the oracle's own orc_point_scalarmul */
ORC_API void orc_bench_extern_scalarmul(void (*fn)(void *, const void *, const void *), const orc_point *base,
                                        const orc_scalar *s, size_t n_in, int nsamples, int ntests, double *times);
ORC_API void orc_point_encode_batch(uint8_t *out56, const orc_point *p, size_t n, int nthreads);
/* msgs: n fixed-length messages of msglen bytes each, contiguous. status[i] = -1/0 */
ORC_API void orc_ed448_verify_batch(int32_t *status, const uint8_t *sig114, const uint8_t *pk57,
                                    const uint8_t *msgs, size_t msglen, uint8_t prehashed,
                                    const uint8_t *ctx, uint8_t ctxlen, size_t n, int nthreads);
ORC_API void orc_ed448_sign_batch(uint8_t *sig114, const uint8_t *sk57, const uint8_t *pk57,
                                  const uint8_t *msgs, size_t msglen, uint8_t prehashed,
                                  const uint8_t *ctx, uint8_t ctxlen, size_t n, int nthreads);
ORC_API void orc_ed448_derive_public_key_batch(uint8_t *pk57, const uint8_t *sk57, size_t n, int nthreads);
ORC_API void orc_point_double_scalarmul_batch(orc_point *out, const orc_point *b1, const orc_scalar *s1, const orc_point *b2,
                                              const orc_scalar *s2, size_t n, int nthreads);
ORC_API void orc_point_dual_scalarmul_batch(orc_point *out1, orc_point *out2, const orc_point *b, const orc_scalar *s1,
                                            const orc_scalar *s2, size_t n, int nthreads);
ORC_API void orc_direct_scalarmul_batch(uint8_t *out56, int32_t *status, const uint8_t *base56, const orc_scalar *s,
                                        int allow_identity, int short_circuit, size_t n, int nthreads);
ORC_API void orc_base_table_entries(uint8_t *out168, unsigned bits, size_t first, size_t count, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
