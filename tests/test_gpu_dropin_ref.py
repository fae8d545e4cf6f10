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
