/* dropin_test_suite_tail.cxx -- TEST INFRASTRUCTURE (oracle/Makefile target `dropin`).  Appended to the reference's own
 * test/test_goldilocks.cxx in one translation unit (its tests are static members of a template): runs the parts of the
 * reference's suite that exercise scalar, point, EdDSA and X448 code -- all of which resolves to libgoldilocks_amd.so here;
 * SHAKE and the deterministic RNG the tests draw from are the reference's own src/shake.c, spongerng.c.
 * Left out: test_elligator (needs the inverse maps, SURVEY section 2 #11: out of scope). */
#undef main
int main() {
    typedef Tests<Ed448Goldilocks> T;
    printf("Testing %s through libgoldilocks_amd, %ld iterations per loop:\n", Ed448Goldilocks::name(), (long)NTESTS);
    T::test_arithmetic();
    T::test_ec();
    T::test_eddsa();
    T::test_x448();
    T::test_convert_eddsa_to_x();
    T::test_cfrg_crypto();
    T::test_cfrg_vectors();
    T::test_dalek_vectors();
    if (passing) printf("Passed all tests.\n");
    return passing ? 0 : 1;
}
