/* dropin_eddsa_main.c -- TEST INFRASTRUCTURE.  A KAT driver for the reference's OWN EdDSA layer linked
 * against libgoldilocks_amd.so instead of the reference's goldilocks.c (oracle/Makefile target
 * `dropin`): the functions called below are the reference's src/eddsa.c, compiled from
 * /root/reference where it lies; every goldilocks_448_* point function that eddsa.c binds
 * (src/eddsa.c:137, :201, :299: precomputed_scalarmul, mul_by_ratio_and_encode_like_eddsa,
 * decode_like_eddsa_and_mul_by_ratio, base_double_scalarmul_non_secret, point_eq, point_destroy,
 * goldilocks_448_precomputed_base) resolves to this repository's GPU library.  If the exported set
 * were incomplete the link would fail with undefined symbols.
 *
 * usage: dropin_eddsa <sk hex 57> <msg hex | -> <ctx hex | -> <prehashed 0|1>
 * prints:  pk=<hex>\n sig=<hex>\n verify=<-1|0>\n verify_bad=<-1|0>                                */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <goldilocks.h>
#include <goldilocks/ed448.h>

static size_t unhex(uint8_t *out, const char *hex) {
    size_t n = strcmp(hex, "-") ? strlen(hex) / 2 : 0;
    for (size_t i = 0; i < n; i++) {
        unsigned v;
        sscanf(hex + 2 * i, "%2x", &v);
        out[i] = (uint8_t)v;
    }
    return n;
}
static void puthex(const char *name, const uint8_t *b, size_t n) {
    printf("%s=", name);
    for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
    printf("\n");
}
int main(int argc, char **argv) {
    if (argc != 5) return 2;
    static uint8_t sk[57], pk[57], sig[114], msg[4096], ctx[256];
    if (unhex(sk, argv[1]) != 57 || strlen(argv[2]) / 2 > sizeof msg || strlen(argv[3]) / 2 > 255) return 2;
    size_t mlen = unhex(msg, argv[2]), clen = unhex(ctx, argv[3]);
    uint8_t ph = (uint8_t)atoi(argv[4]);
    goldilocks_ed448_derive_public_key(pk, sk);                        /* src/eddsa.c:98-147 */
    goldilocks_ed448_sign(sig, sk, pk, msg, mlen, ph, ctx, (uint8_t)clen);   /* src/eddsa.c:149-230 */
    puthex("pk", pk, 57);
    puthex("sig", sig, 114);
    printf("verify=%d\n", (int)goldilocks_ed448_verify(sig, pk, msg, mlen, ph, ctx, (uint8_t)clen));  /* :253-306 */
    sig[60] ^= 4;
    printf("verify_bad=%d\n", (int)goldilocks_ed448_verify(sig, pk, msg, mlen, ph, ctx, (uint8_t)clen));
    return 0;
}
