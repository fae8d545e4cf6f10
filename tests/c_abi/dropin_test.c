/* dropin_test.c -- a plain C99 caller written the way a user of the reference library writes code
 * (stack-allocated goldilocks_448_point_p / scalar_p, the reference's function names), compiled
 * against include/goldilocks_amd.h and linked with -lgoldilocks_amd.  Run by the GPU test
 * tests/test_gpu_c_abi.py; exits 0 on success.
 *
 * Checks: RFC 8032 Ed448 test vector 1 (blank message) verifies and a corrupted copy does not;
 * k*B through point_scalarmul, precomputed_scalarmul and repeated point_add agree byte for byte
 * (the reference's test_dalek_vectors idea); sign(derive_public_key) round-trips; X448 DH agrees. */
#include <stdio.h>
#include <string.h>
#include "goldilocks_amd.h"

static int hexval(char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }
static void unhex(uint8_t *out, const char *hex, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(hexval(hex[2 * i]) << 4 | hexval(hex[2 * i + 1]));
}

int main(void) {
    /* RFC 8032 section 7.4, "Blank" */
    static const char *PK = "5fd7449b59b461fd2ce787ec616ad46a1da1342485a70e1f8a0ea75d80e96778"
                            "edf124769b46c7061bd6783df1e50f6cd1fa1abeafe8256180";
    static const char *SIG = "533a37f6bbe457251f023c0d88f976ae2dfb504a843e34d2074fd823d41a591f"
                             "2b233f034f628281f2fd7a22ddd47d7828c59bd0a21bfd3980ff0d2028d4b18a"
                             "9df63e006c5d1c2d345b925d8dc00b4104852db99ac5c7cdda8530a113a0f4db"
                             "b61149f05a7363268c71d95808ff2e652600";
    uint8_t pk[57], sig[114], ser1[56], ser2[56], ser3[56];
    unhex(pk, PK, 57);
    unhex(sig, SIG, 114);
    if (goldilocks_ed448_verify(sig, pk, (const uint8_t *)"", 0, 0, (const uint8_t *)"", 0) != GOLDILOCKS_SUCCESS) {
        puts("RFC 8032 vector 1 rejected");
        return 1;
    }
    sig[20] ^= 1;
    if (goldilocks_ed448_verify(sig, pk, (const uint8_t *)"", 0, 0, (const uint8_t *)"", 0) != GOLDILOCKS_FAILURE) {
        puts("corrupted signature accepted");
        return 1;
    }

    goldilocks_448_point_p acc, via_mul, via_comb;
    goldilocks_448_scalar_p k;
    memcpy(acc, goldilocks_448_point_identity, sizeof(acc));
    for (unsigned i = 0; i < 8; i++) {
        memset(k, 0, sizeof(k));
        k->limb[0] = i;
        goldilocks_448_point_scalarmul(via_mul, goldilocks_448_point_base, k);
        goldilocks_448_precomputed_scalarmul(via_comb, goldilocks_448_precomputed_base, k);
        goldilocks_448_point_encode(ser1, acc);
        goldilocks_448_point_encode(ser2, via_mul);
        goldilocks_448_point_encode(ser3, via_comb);
        if (memcmp(ser1, ser2, 56) || memcmp(ser1, ser3, 56)) {
            printf("%u*B disagrees between add chain, scalarmul and comb\n", i);
            return 1;
        }
        if (!goldilocks_448_point_valid(via_mul) || !goldilocks_448_point_eq(via_mul, acc)) {
            puts("point_valid / point_eq");
            return 1;
        }
        goldilocks_448_point_add(acc, acc, goldilocks_448_point_base);   /* output aliases an input */
    }

    uint8_t sk[57], mypk[57], mysig[114];
    for (int i = 0; i < 57; i++) sk[i] = (uint8_t)(7 * i + 1);
    goldilocks_ed448_derive_public_key(mypk, sk);
    goldilocks_ed448_sign(mysig, sk, mypk, (const uint8_t *)"drop-in", 7, 0, (const uint8_t *)"ctx", 3);
    if (goldilocks_ed448_verify(mysig, mypk, (const uint8_t *)"drop-in", 7, 0, (const uint8_t *)"ctx", 3) !=
        GOLDILOCKS_SUCCESS) {
        puts("sign/verify round trip");
        return 1;
    }

    uint8_t a[56], b[56], pa[56], pb[56], s1[56], s2[56];
    for (int i = 0; i < 56; i++) { a[i] = (uint8_t)(3 * i + 5); b[i] = (uint8_t)(11 * i + 2); }
    goldilocks_x448_derive_public_key(pa, a);
    goldilocks_x448_derive_public_key(pb, b);
    if (goldilocks_x448(s1, pb, a) != GOLDILOCKS_SUCCESS || goldilocks_x448(s2, pa, b) != GOLDILOCKS_SUCCESS ||
        memcmp(s1, s2, 56)) {
        puts("x448 DH");
        return 1;
    }

    /* the scalar API (round 6): (x * y) / y == x, x - x == 0, 2 * (x / 2) == x, decode / encode, and a point identity through it:
       (x + y) * B == x * B + y * B */
    {
        goldilocks_448_scalar_p x, y, t, u;
        uint8_t ser[56], back[56];
        goldilocks_448_point_p xb, yb, sum, direct;
        for (int i = 0; i < 56; i++) ser[i] = (uint8_t)(29 * i + 3);
        ser[55] &= 0x3f;                                     /* below q */
        if (goldilocks_448_scalar_decode(x, ser) != GOLDILOCKS_SUCCESS) { puts("scalar_decode"); return 1; }
        goldilocks_448_scalar_encode(back, x);
        if (memcmp(ser, back, 56)) { puts("scalar_encode"); return 1; }
        goldilocks_448_scalar_decode_long(y, (const unsigned char *)"a long string of more than fifty-six bytes, reduced modulo the group order", 73);
        goldilocks_448_scalar_mul(t, x, y);
        if (goldilocks_448_scalar_invert(u, y) != GOLDILOCKS_SUCCESS) { puts("scalar_invert"); return 1; }
        goldilocks_448_scalar_mul(t, t, u);                  /* in place */
        if (!goldilocks_448_scalar_eq(t, x)) { puts("(x * y) / y != x"); return 1; }
        goldilocks_448_scalar_sub(t, x, x);
        if (!goldilocks_448_scalar_eq(t, goldilocks_448_scalar_zero) || goldilocks_448_scalar_invert(u, t) != GOLDILOCKS_FAILURE) {
            puts("x - x, 1 / 0");
            return 1;
        }
        goldilocks_448_scalar_halve(t, x);
        goldilocks_448_scalar_add(t, t, t);
        if (!goldilocks_448_scalar_eq(t, x)) { puts("2 * (x / 2) != x"); return 1; }
        goldilocks_448_scalar_set_unsigned(u, 5);
        goldilocks_448_scalar_cond_sel(t, x, u, 1);
        if (!goldilocks_448_scalar_eq(t, u) || u->limb[0] != 5) { puts("set_unsigned / cond_sel"); return 1; }
        goldilocks_448_scalar_add(t, x, y);
        goldilocks_448_point_scalarmul(xb, goldilocks_448_point_base, x);
        goldilocks_448_point_scalarmul(yb, goldilocks_448_point_base, y);
        goldilocks_448_point_add(sum, xb, yb);
        goldilocks_448_precomputed_scalarmul(direct, goldilocks_448_precomputed_base, t);
        if (!goldilocks_448_point_eq(sum, direct)) { puts("(x + y) * B != x * B + y * B"); return 1; }
        goldilocks_448_point_debugging_torque(sum, sum);
        goldilocks_448_point_debugging_pscale(sum, sum, ser);
        goldilocks_448_point_encode(ser1, sum);
        goldilocks_448_point_encode(ser2, direct);
        if (memcmp(ser1, ser2, 56)) { puts("torque / pscale changed the encoding"); return 1; }
        goldilocks_448_scalar_destroy(x);
        if (!goldilocks_448_scalar_eq(x, goldilocks_448_scalar_zero)) { puts("scalar_destroy"); return 1; }
    }

    /* batch entry points: host arrays, then the same batch sharded over "two GPUs" (device 0 listed
       twice on a one-GPU box) and with index-independent table access; all must agree */
    enum { NB = 300 };
    static goldilocks_448_scalar_s ks[NB];
    static goldilocks_448_point_s fixed1[NB], fixed2[NB], var1[NB], var2[NB];
    for (int i = 0; i < NB; i++) {
        memset(&ks[i], 0, sizeof(ks[i]));
        ks[i].limb[0] = 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1);
        ks[i].limb[3] = (uint64_t)i * i + 17;
    }
    if (goldilocks_448_precomputed_scalarmul_batch(fixed1, goldilocks_448_precomputed_base, ks, NB) ||
        goldilocks_448_point_scalarmul_batch(var1, fixed1, ks, NB)) {
        printf("batch: %s\n", goldilocks_amd_last_error());
        return 1;
    }
    const int two[2] = {0, 0};
    if (goldilocks_amd_use_devices(two, 2) || goldilocks_amd_set_table_access(GOLDILOCKS_AMD_TABLES_INDEX_INDEPENDENT) ||
        goldilocks_448_precomputed_scalarmul_batch(fixed2, goldilocks_448_precomputed_base, ks, NB) ||
        goldilocks_448_point_scalarmul_batch(var2, fixed2, ks, NB) || goldilocks_amd_use_devices(NULL, 0) ||
        goldilocks_amd_set_table_access(GOLDILOCKS_AMD_TABLES_FAST)) {
        printf("sharded batch: %s\n", goldilocks_amd_last_error());
        return 1;
    }
    for (int i = 0; i < NB; i++) {
        if (!goldilocks_448_point_eq(&fixed1[i], &fixed2[i]) || !goldilocks_448_point_eq(&var1[i], &var2[i])) {
            printf("sharded / index-independent batch differs at %d\n", i);
            return 1;
        }
    }
    puts("dropin_test ok");
    return 0;
}
