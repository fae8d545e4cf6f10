"""CPU tests, build container only: live differential run of the oracle against the REAL reference
(oracle/_ref, arch_ref64).  Raw limbs must agree, not just encodings.  Skipped where the reference
build is absent (the GPU box has the prebuilt .so; a bare checkout has not)."""
import ctypes as C
import random

import pytest

from _libs import Point, Scalar, buf, have_ref, ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (make -C oracle ref)")


def test_tables_identical(O):
    R = ref()
    assert bytes(O.orc_point_base().contents) == bytes(R.point_base)
    assert bytes(O.orc_precomputed_base().contents) == C.string_at(R.precomputed_base.value, 15360)


def test_differential(O):
    R = ref()
    rnd = random.Random(11)
    rb = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    for it in range(60):
        p1, p2, s, t, s2 = Point(), Point(), Scalar(), Scalar(), Scalar()
        R.goldilocks_448_point_from_hash_uniform(C.byref(p1), buf(rb(112)))
        R.goldilocks_448_point_from_hash_uniform(C.byref(p2), buf(rb(112)))
        raw = rb(72)
        R.goldilocks_448_scalar_decode_long(C.byref(s), buf(raw), 72)
        O.orc_scalar_decode_long(C.byref(s2), buf(raw), 72)
        assert bytes(s) == bytes(s2)
        R.goldilocks_448_scalar_decode_long(C.byref(t), buf(rb(72)), 72)
        a, b = Point(), Point()
        R.goldilocks_448_point_scalarmul(C.byref(a), C.byref(p1), C.byref(s))
        O.orc_point_scalarmul(C.byref(b), C.byref(p1), C.byref(s))
        assert bytes(a) == bytes(b)
        R.goldilocks_448_precomputed_scalarmul(C.byref(a), R.precomputed_base, C.byref(s))
        O.orc_precomputed_scalarmul(C.byref(b), O.orc_precomputed_base(), C.byref(s))
        assert bytes(a) == bytes(b)
        R.goldilocks_448_point_double_scalarmul(C.byref(a), C.byref(p1), C.byref(s), C.byref(p2), C.byref(t))
        O.orc_point_double_scalarmul(C.byref(b), C.byref(p1), C.byref(s), C.byref(p2), C.byref(t))
        assert bytes(a) == bytes(b)
        R.goldilocks_448_base_double_scalarmul_non_secret(C.byref(a), C.byref(s), C.byref(p2), C.byref(t))
        O.orc_base_double_scalarmul_non_secret(C.byref(b), C.byref(s), C.byref(p2), C.byref(t))
        assert bytes(a) == bytes(b)
        e1, e2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
        R.goldilocks_448_point_encode(e1, C.byref(a)); O.orc_point_encode(e2, C.byref(a))
        assert bytes(e1) == bytes(e2)
        d1, d2 = Point(), Point()
        assert R.goldilocks_448_point_decode(C.byref(d1), e1, 0) == O.orc_point_decode(C.byref(d2), e1, 0) == -1
        assert bytes(d1) == bytes(d2)
        g = rb(56)
        assert R.goldilocks_448_point_decode(C.byref(d1), buf(g), 0) == O.orc_point_decode(C.byref(d2), buf(g), 0)
        f1, f2 = (C.c_uint8 * 57)(), (C.c_uint8 * 57)()
        R.goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa(f1, C.byref(a)); O.orc_point_encode_like_eddsa(f2, C.byref(a))
        assert bytes(f1) == bytes(f2)
        r1 = R.goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio(C.byref(d1), f1)
        assert r1 == O.orc_point_decode_like_eddsa(C.byref(d2), f1) and bytes(d1) == bytes(d2)
        sk, msg, ctx = rb(57), rb(it * 3), rb(it % 7)
        pk1, pk2, s1, s2_ = (C.c_uint8 * 57)(), (C.c_uint8 * 57)(), (C.c_uint8 * 114)(), (C.c_uint8 * 114)()
        R.goldilocks_ed448_derive_public_key(pk1, buf(sk)); O.orc_ed448_derive_public_key(pk2, buf(sk))
        assert bytes(pk1) == bytes(pk2)
        m, c = (buf(msg) if msg else None), (buf(ctx) if ctx else None)
        R.goldilocks_ed448_sign(s1, buf(sk), pk1, m, len(msg), it & 1, c, len(ctx))
        O.orc_ed448_sign(s2_, buf(sk), pk1, m, len(msg), it & 1, c, len(ctx))
        assert bytes(s1) == bytes(s2_)
        assert R.goldilocks_ed448_verify(s1, pk1, m, len(msg), it & 1, c, len(ctx)) == -1
        assert O.orc_ed448_verify(s1, pk1, m, len(msg), it & 1, c, len(ctx)) == -1
        bad = bytearray(s1); bad[rnd.randrange(114)] ^= 1 << rnd.randrange(8)
        assert R.goldilocks_ed448_verify(buf(bad), pk1, m, len(msg), it & 1, c, len(ctx)) == \
            O.orc_ed448_verify(buf(bad), pk1, m, len(msg), it & 1, c, len(ctx))


def test_differential_ten_thousand_scalar_multiplications(O):
    """The hot path at volume: 10 000 random (point, scalar) pairs through the oracle and the real reference -- variable
    base, fixed base (raw limbs of both), decaf encoding; every tenth pair also the two-base multiplication and an
    EdDSA verification of a signature with one flipped bit somewhere.  (The 60 iterations above cover every entry
    point; this one hunts rare carries in the arithmetic the GPU tests are checked against.)"""
    R = ref()
    rnd = random.Random(2024)
    rb = lambda n: rnd.getrandbits(8 * n).to_bytes(n, "little")
    p1, p2, s, t, s2, a, b = Point(), Point(), Scalar(), Scalar(), Scalar(), Point(), Point()
    e1, e2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
    for it in range(10000):
        R.goldilocks_448_point_from_hash_uniform(C.byref(p1), buf(rb(112)))
        raw = rb(72) if it % 50 else bytes([0xff]) * 72            # now and then the largest input
        R.goldilocks_448_scalar_decode_long(C.byref(s), buf(raw), 72)
        O.orc_scalar_decode_long(C.byref(s2), buf(raw), 72)
        assert bytes(s) == bytes(s2), it
        R.goldilocks_448_point_scalarmul(C.byref(a), C.byref(p1), C.byref(s))
        O.orc_point_scalarmul(C.byref(b), C.byref(p1), C.byref(s))
        assert bytes(a) == bytes(b), it
        R.goldilocks_448_point_encode(e1, C.byref(a)); O.orc_point_encode(e2, C.byref(b))
        assert bytes(e1) == bytes(e2), it
        R.goldilocks_448_precomputed_scalarmul(C.byref(a), R.precomputed_base, C.byref(s))
        O.orc_precomputed_scalarmul(C.byref(b), O.orc_precomputed_base(), C.byref(s))
        assert bytes(a) == bytes(b), it
        if it % 10:
            continue
        R.goldilocks_448_point_from_hash_uniform(C.byref(p2), buf(rb(112)))
        R.goldilocks_448_scalar_decode_long(C.byref(t), buf(rb(72)), 72)
        R.goldilocks_448_point_double_scalarmul(C.byref(a), C.byref(p1), C.byref(s), C.byref(p2), C.byref(t))
        O.orc_point_double_scalarmul(C.byref(b), C.byref(p1), C.byref(s), C.byref(p2), C.byref(t))
        assert bytes(a) == bytes(b), it
        sk, msg = rb(57), rb(it % 200)
        pk, sg = (C.c_uint8 * 57)(), (C.c_uint8 * 114)()
        R.goldilocks_ed448_derive_public_key(pk, buf(sk))
        m = buf(msg) if msg else None
        R.goldilocks_ed448_sign(sg, buf(sk), pk, m, len(msg), 0, None, 0)
        assert O.orc_ed448_verify(sg, pk, m, len(msg), 0, None, 0) == -1, it
        bad = bytearray(sg); bad[rnd.randrange(114)] ^= 1 << rnd.randrange(8)
        assert R.goldilocks_ed448_verify(buf(bad), pk, m, len(msg), 0, None, 0) == \
            O.orc_ed448_verify(buf(bad), pk, m, len(msg), 0, None, 0), it


def test_decode_special_encodings_differential(O):
    """Decaf and EdDSA decoding of hand-picked encodings (identity, 1, p-1, values >= p, small even and
    odd values, set high bits): status and, when accepted, the raw output limbs match the reference."""
    from _libs import P
    R = ref()
    vals = [0, 1, 2, 3, 4, 5, P - 1, P - 2, P, P + 1, 2**447, 2**448 - 1, (P - 1) // 2, (P + 1) // 2]
    for v in vals:
        enc = v.to_bytes(56, "little")
        for allow in (0, 1):
            d1, d2 = Point(), Point()
            r1 = R.goldilocks_448_point_decode(C.byref(d1), buf(enc), allow)
            r2 = O.orc_point_decode(C.byref(d2), buf(enc), allow)
            assert r1 == r2, (v, allow)
            if r1 == -1:
                assert bytes(d1) == bytes(d2), (v, allow)
        for last in (0x00, 0x80, 0x01, 0x7f):
            e57 = enc + bytes([last])
            d1, d2 = Point(), Point()
            r1 = R.goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio(C.byref(d1), buf(e57))
            r2 = O.orc_point_decode_like_eddsa(C.byref(d2), buf(e57))
            assert r1 == r2, (v, last)
            if r1 == -1:
                assert bytes(d1) == bytes(d2), (v, last)


def test_precompute_matches_reference(O):
    from _libs import Precomputed
    R = ref()
    p = Point()
    R.goldilocks_448_point_from_hash_uniform(C.byref(p), buf(bytes(range(112))))
    t1, t2 = (C.c_uint8 * 15360)(), Precomputed()
    R.goldilocks_448_precompute(t1, C.byref(p))
    O.orc_precompute(C.byref(t2), C.byref(p))
    assert bytes(t1) == bytes(t2)


def test_x448_differential(O):
    R = ref()
    rnd = random.Random(12)
    from _libs import P
    special = [x.to_bytes(56, "little") for x in (0, 1, P - 1, P, P + 1, 2**448 - 1, 2, 5)]   # low order / >= p
    for it in range(100):
        b = bytes(rnd.getrandbits(8) for _ in range(56)) if it >= len(special) else special[it]
        s = bytes(rnd.getrandbits(8) for _ in range(56))
        o1, o2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
        assert R.goldilocks_x448(o1, buf(b), buf(s)) == O.orc_x448(o2, buf(b), buf(s)) and bytes(o1) == bytes(o2)
        R.goldilocks_x448_derive_public_key(o1, buf(s)); O.orc_x448_derive_public_key(o2, buf(s))
        assert bytes(o1) == bytes(o2)


def test_x448_conversions_differential(O):
    """goldilocks_ed448_convert_public_key_to_x448 / _private_key_to_x448 / _derive_secret_scalar and
    goldilocks_448_point_mul_by_ratio_and_encode_like_x448 (src/goldilocks.c:1079-1115, src/eddsa.c:83-128): the oracle's
    restatements against the reference compiled here; and the reference's own test of them (test_goldilocks.cxx:625-655:
    the public key converted == the X448 public key of the private key converted)."""
    from _libs import P
    R = ref()
    rnd = random.Random(14)
    special = [x.to_bytes(56, "little") + b"\x80" for x in (0, 1, P - 1, P, P + 1, 2**448 - 1)]   # y = +-1: 1/(1 - y^2) = 1/0; y >= p
    for it in range(100):
        sk = bytes(rnd.getrandbits(8) for _ in range(57))
        pk = (C.c_uint8 * 57)()
        R.goldilocks_ed448_derive_public_key(pk, buf(sk))
        ed = special[it] if it < len(special) else bytes(pk)
        o1, o2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
        R.goldilocks_ed448_convert_public_key_to_x448(o1, buf(ed)); O.orc_ed448_convert_public_key_to_x448(o2, buf(ed))
        assert bytes(o1) == bytes(o2), it
        x1, x2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
        R.goldilocks_ed448_convert_private_key_to_x448(x1, buf(sk)); O.orc_ed448_convert_private_key_to_x448(x2, buf(sk))
        assert bytes(x1) == bytes(x2)
        if it >= len(special):
            R.goldilocks_x448_derive_public_key(x2, x1)
            assert bytes(x2) == bytes(o1)
        s1, s2 = Scalar(), Scalar()
        R.goldilocks_ed448_derive_secret_scalar(C.byref(s1), buf(sk)); O.orc_ed448_derive_secret_scalar(C.byref(s2), buf(sk))
        assert bytes(s1) == bytes(s2)
        p = Point()
        if it == 0:
            C.memmove(C.byref(p), R_identity(R), 256)          # x = 0: 1/x = 0
        else:
            R.goldilocks_448_point_from_hash_uniform(C.byref(p), buf(bytes(rnd.getrandbits(8) for _ in range(112))))
        R.goldilocks_448_point_mul_by_ratio_and_encode_like_x448(o1, C.byref(p)); O.orc_point_encode_like_x448(o2, C.byref(p))
        assert bytes(o1) == bytes(o2), it


def test_scalar_arithmetic_differential(O):
    """src/scalar.c:30-332 as an API: add, sub, mul, halve, invert, decode (56 bytes, range check) and decode_long at
    lengths 0 ... 200 -- the oracle's restatements against the reference compiled here, on random scalars and 0, 1, q - 1."""
    from _libs import Q
    R = ref()
    rnd = random.Random(16)
    def sc(x):
        s = Scalar()
        C.memmove(C.byref(s), (x % Q).to_bytes(56, "little"), 56)
        return s
    edge = [0, 1, 2, Q - 1, Q - 2, (Q + 1) // 2]
    for it in range(120):
        x = edge[it % len(edge)] if it < 12 else rnd.getrandbits(446) % Q
        y = edge[(it // len(edge)) % len(edge)] if it < 36 else rnd.getrandbits(446) % Q
        a, b, o1, o2 = sc(x), sc(y), Scalar(), Scalar()
        for name in ("add", "sub", "mul"):
            getattr(R, "goldilocks_448_scalar_" + name)(C.byref(o1), C.byref(a), C.byref(b))
            getattr(O, "orc_scalar_" + name)(C.byref(o2), C.byref(a), C.byref(b))
            assert bytes(o1) == bytes(o2) == ({"add": x + y, "sub": x - y, "mul": x * y}[name] % Q).to_bytes(56, "little"), (name, it)
        R.goldilocks_448_scalar_halve(C.byref(o1), C.byref(a)); O.orc_scalar_halve(C.byref(o2), C.byref(a))
        assert bytes(o1) == bytes(o2) == (x * pow(2, -1, Q) % Q).to_bytes(56, "little")
        if it % 6 == 0:
            r1 = R.goldilocks_448_scalar_invert(C.byref(o1), C.byref(a)); r2 = O.orc_scalar_invert(C.byref(o2), C.byref(a))
            assert bytes(o1) == bytes(o2) == (pow(x, -1, Q) if x else 0).to_bytes(56, "little") and r1 == r2 == (-1 if x else 0)
        raw = [0, Q - 1, Q, Q + 1, 2**448 - 1][it] if it < 5 else rnd.getrandbits(448)
        ser = raw.to_bytes(56, "little")
        r1 = R.goldilocks_448_scalar_decode(C.byref(o1), buf(ser)); r2 = O.orc_scalar_decode(C.byref(o2), buf(ser))
        assert bytes(o1) == bytes(o2) == (raw % Q).to_bytes(56, "little") and r1 == r2 == (-1 if raw < Q else 0)
        n = it if it < 80 else rnd.randrange(80, 201)
        long = bytes(rnd.getrandbits(8) for _ in range(n)) if it != 57 else b"\xff" * 57
        R.goldilocks_448_scalar_decode_long(C.byref(o1), buf(long + b"\0"), n); O.orc_scalar_decode_long(C.byref(o2), buf(long + b"\0"), n)
        assert bytes(o1) == bytes(o2) == (int.from_bytes(long, "little") % Q).to_bytes(56, "little"), n


def test_debugging_helpers_differential(O):
    """goldilocks_448_point_debugging_torque / _pscale (src/goldilocks.c:675-701): raw limbs of the oracle's restatement
    against the reference compiled here -- random factors, 0 (counts as 1), p (reads as 0 too), values beyond p."""
    from _libs import P
    R = ref()
    rnd = random.Random(15)
    factors = [x.to_bytes(56, "little") for x in (0, 1, P - 1, P, P + 1, 2**448 - 1)]
    for it in range(60):
        p, a, b = Point(), Point(), Point()
        R.goldilocks_448_point_from_hash_uniform(C.byref(p), buf(bytes(rnd.getrandbits(8) for _ in range(112))))
        R.goldilocks_448_point_debugging_torque(C.byref(a), C.byref(p)); O.orc_point_debugging_torque(C.byref(b), C.byref(p))
        assert bytes(a) == bytes(b)
        f = factors[it] if it < len(factors) else bytes(rnd.getrandbits(8) for _ in range(56))
        R.goldilocks_448_point_debugging_pscale(C.byref(a), C.byref(p), buf(f)); O.orc_point_debugging_pscale(C.byref(b), C.byref(p), buf(f))
        assert bytes(a) == bytes(b), it
        assert R.goldilocks_448_point_eq(C.byref(a), C.byref(p)) and R.goldilocks_448_point_valid(C.byref(a))


def R_identity(R):
    return C.addressof(Point.in_dll(R, "goldilocks_448_point_identity"))


def test_elligator_differential(O):
    R = ref()
    rnd = random.Random(13)
    for it in range(100):
        h = bytes(rnd.getrandbits(8) for _ in range(112)) if it > 1 else (bytes(112), b"\xff" * 112)[it]
        a, b = Point(), Point()
        R.goldilocks_448_point_from_hash_nonuniform(C.byref(a), buf(h)); O.orc_point_from_hash_nonuniform(C.byref(b), buf(h))
        assert bytes(a) == bytes(b)
        R.goldilocks_448_point_from_hash_uniform(C.byref(a), buf(h)); O.orc_point_from_hash_uniform(C.byref(b), buf(h))
        assert bytes(a) == bytes(b)
