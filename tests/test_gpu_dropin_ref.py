"""The drop-in boundary proved with the reference's OWN caller: oracle/_ref/dropin_eddsa is the
reference's src/eddsa.c (+ scalar.c, shake.c, utils.c) compiled from /root/reference in the build
container and linked against libgoldilocks_amd.so in place of the reference's goldilocks.c and field
backend (oracle/Makefile target `dropin`).  Every point operation eddsa.c
makes (src/eddsa.c:137, :201, :299) therefore runs on the GPU through the reference's exact symbol
names.  RFC 8032's Ed448 vectors must come out byte for byte."""
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "dropin_eddsa")
needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/dropin_eddsa is built where /root/reference exists")


@needs_exe
def test_reference_caller_binds_only_exported_symbols():
    """CPU: what the reference's eddsa.c leaves undefined is exactly a subset of what the header declares."""
    import libgoldilocks_amd as ga
    out = subprocess.check_output(["nm", "-u", EXE], text=True)
    wanted = {l.split()[-1].split("@")[0] for l in out.splitlines() if "goldilocks" in l}
    assert {"goldilocks_448_precomputed_scalarmul", "goldilocks_448_base_double_scalarmul_non_secret",
            "goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio", "goldilocks_448_point_eq",
            "goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa", "goldilocks_448_point_destroy"} <= wanted
    assert wanted <= set(ga.FUNCTIONS) | set(ga.DATA_SYMBOLS)


@needs_exe
@pytest.mark.gpu
def test_rfc8032_through_the_references_eddsa_layer():
    kats = json.load(open(os.path.join(HERE, "golden", "kats.json")))["rfc8032_ed448"]
    assert len(kats) == 11
    for c in kats:
        msg = bytes.fromhex(c["message"])
        if c["prehashed"]:
            msg = hashlib.shake_256(msg).digest(64)
        r = subprocess.run([EXE, c["sk"], msg.hex() or "-", c["context"] or "-", "1" if c["prehashed"] else "0"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = dict(l.split("=", 1) for l in r.stdout.split())
        assert got["pk"] == c["pk"] and got["sig"] == c["sig"], c["message"][:16]
        assert got["verify"] == "-1" and got["verify_bad"] == "0"


SUITE = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "dropin_test_suite")
needs_suite = pytest.mark.skipif(not os.path.exists(SUITE), reason="oracle/_ref/dropin_test_suite is built where /root/reference exists")


@needs_suite
def test_reference_test_suite_binds_the_library_for_everything_it_tests():
    """CPU: the reference's test/test_goldilocks.cxx, as linked by oracle/Makefile, takes every point, EdDSA and X448
    function its tests call from libgoldilocks_amd.so -- sign, verify and derive_public_key included (the reference's
    eddsa.c is on the link line for its prehash glue only, its other symbols made local) -- and nothing it needs is
    missing from the header."""
    import libgoldilocks_amd as ga
    out = subprocess.check_output(["nm", "-u", SUITE], text=True)
    wanted = {l.split()[-1].split("@")[0] for l in out.splitlines() if "goldilocks" in l}
    assert {"goldilocks_ed448_sign", "goldilocks_ed448_verify", "goldilocks_ed448_derive_public_key", "goldilocks_x448",
            "goldilocks_x448_derive_public_key", "goldilocks_448_point_scalarmul", "goldilocks_448_point_double_scalarmul",
            "goldilocks_448_point_dual_scalarmul", "goldilocks_448_direct_scalarmul", "goldilocks_448_precompute",
            "goldilocks_448_point_debugging_torque", "goldilocks_448_point_debugging_pscale",
            "goldilocks_ed448_convert_public_key_to_x448", "goldilocks_ed448_convert_private_key_to_x448",
            "goldilocks_448_point_from_hash_uniform", "goldilocks_448_point_mul_by_ratio_and_encode_like_x448",
            "goldilocks_448_scalar_add", "goldilocks_448_scalar_sub", "goldilocks_448_scalar_mul", "goldilocks_448_scalar_invert",
            "goldilocks_448_scalar_halve", "goldilocks_448_scalar_decode_long", "goldilocks_448_scalar_eq"} <= wanted
    assert wanted <= set(ga.FUNCTIONS) | set(ga.DATA_SYMBOLS), wanted - set(ga.FUNCTIONS) - set(ga.DATA_SYMBOLS)
    defined = subprocess.check_output(["nm", "--defined-only", "-g", SUITE], text=True)
    assert "goldilocks_ed448_sign\n" not in defined and "goldilocks_448_point_add" not in defined


@needs_suite
@pytest.mark.gpu
def test_reference_test_suite_passes_on_the_library():
    """The reference's OWN property suite (test/test_goldilocks.cxx: test_arithmetic -- the scalars' ring laws and inversion --,
    :316-437 test_ec -- round trips, torque and projective
    scaling, commutativity, associativity, distributivity, double / dual / precomputed / direct multiplications, the
    Elligator sum, the EdDSA encoding round trip --, test_eddsa, test_x448, test_convert_eddsa_to_x, test_cfrg_crypto,
    test_cfrg_vectors with RFC 7748's iterated ladder and RFC 8032's vectors, test_dalek_vectors) with every scalar, point,
    EdDSA and X448 call served by this library, one GPU call each.  Here with 150 iterations per loop; the full 10 000 of the
    reference's source: `oracle/_ref/dropin_test_suite` without the variable (profiles/r06/reference_test_suite.txt)."""
    env = dict(os.environ, GOLDILOCKS_REF_NTESTS="150")
    r = subprocess.run([SUITE], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "Passed all tests." in r.stdout and "FAIL" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for name in ("Arithmetic", "EC", "EdDSA", "X448 Encoding/Decoding", "ECDH using EdDSA keys", "CFRG crypto", "CFRG test vectors", "Test vectors from Dalek"):
        assert name + "..." in r.stdout, name
