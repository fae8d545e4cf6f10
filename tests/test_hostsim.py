"""CPU tests: the device lane arithmetic (libgoldilocks_amd/csrc/*.hpp) compiled for the host with
the GF_CHECKED accumulator -- which traps on any 64-bit accumulator overflow and on violated
subtraction-bias preconditions -- against the oracle.  This exercises the magnitude contract of
gf28.hpp; it is a checker build, never a product path."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import _gen
from _libs import Gf, Point, Scalar, P, Q, buf

HS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")


@pytest.fixture(scope="module")
def H():
    subprocess.check_call(["make", "-s", "-C", HS_DIR])
    return C.CDLL(os.environ.get("GOLDILOCKS_HOSTSIM_LIB") or os.path.join(HS_DIR, "libhostsim.so"))   # or the sanitizer build


def _ser(O, g):
    b = (C.c_uint8 * 56)(); O.orc_gf_serialize(b, C.byref(g)); return bytes(b)


def _enc(O, p):
    b = (C.c_uint8 * 56)(); O.orc_point_encode(b, C.byref(p)); return bytes(b)


def test_field_and_magnitude_contract(H, O):
    rnd = random.Random(2)
    allones = Gf()
    for i in range(8):
        allones.limb[i] = (1 << 56) - 1
    cases = [Gf.from_int(rnd.getrandbits(448) % P) for _ in range(150)] + [allones, Gf.from_int(0), Gf.from_int(P - 1)]
    for a in cases:
        b = cases[rnd.randrange(len(cases))]
        o1, o2 = Gf(), Gf()
        O.orc_gf_mul(C.byref(o1), C.byref(a), C.byref(b)); H.hs_fe_mul(C.byref(o2), C.byref(a), C.byref(b))
        assert _ser(O, o1) == _ser(O, o2)
        O.orc_gf_sqr(C.byref(o1), C.byref(a)); H.hs_fe_sqr(C.byref(o2), C.byref(a))
        assert _ser(O, o1) == _ser(O, o2)
        w = rnd.getrandbits(18)
        O.orc_gf_mulw(C.byref(o1), C.byref(a), w); H.hs_fe_mulw(C.byref(o2), C.byref(a), w)
        assert _ser(O, o1) == _ser(O, o2)
        for ma, mb in ((2, 2), (5, 1), (1, 5), (4, 1), (3, 1), (1, 4), (2, 3), (3, 2), (6, 1)):   # documented limits
            H.hs_fe_mul_mag(C.byref(o2), C.byref(a), C.byref(b), ma, mb)
            assert o2.value() == a.value() * b.value() * ma * mb % P
        H.hs_fe_sqr_mag(C.byref(o2), C.byref(a), 2)
        assert o2.value() == 4 * a.value() ** 2 % P
    H.hs_fe_isr.restype = C.c_int
    for a in cases[:12]:
        o1, o2 = Gf(), Gf()
        m1 = O.orc_gf_isr(C.byref(o1), C.byref(a)); m2 = H.hs_fe_isr(C.byref(o2), C.byref(a))
        assert (m1 != 0) == (m2 != 0) and _ser(O, o1) == _ser(O, o2)
        bs = (C.c_uint8 * 56)(); H.hs_fe_serialize(bs, C.byref(a)); assert bytes(bs) == _ser(O, a)


def _signed_cases(rnd):
    allones = Gf()
    for i in range(8):
        allones.limb[i] = (1 << 56) - 1
    return [Gf.from_int(rnd.getrandbits(448) % P) for _ in range(150)] + [allones, Gf.from_int(0), Gf.from_int(1), Gf.from_int(P - 1)]


def test_signed_paired_field_layer_at_its_limits(H):
    """gf28s.hpp (the ladders' field layer) under the checker: signed 64-bit finished columns, 32-bit pre-added halves,
    pair-wise additions whose low halves must not carry -- at the magnitudes its header documents, with the all-ones
    element (every limb 2^28 - 1) among the operands."""
    rnd = random.Random(5)
    cases = _signed_cases(rnd)
    for a in cases:
        b = cases[rnd.randrange(len(cases))]
        c = cases[rnd.randrange(len(cases))]
        o = Gf()
        for ka, kb in ((1, 1), (2, 1), (1, 2), (3, 1), (2, -1), (-1, 2), (-1, -1), (-2, 1), (1, -2), (3, -1), (-3, 1)):
            H.hs_sfe_mul_mag(C.byref(o), C.byref(a), C.byref(b), ka, kb)
            assert o.value() == a.value() * b.value() * ka * kb % P, (ka, kb)
        for ka, sum2 in ((1, 0), (-1, 0), (1, 1), (2, 1)):
            H.hs_sfe_sqr_mag(C.byref(o), C.byref(a), ka, sum2)
            assert o.value() == (ka * a.value()) ** 2 % P, (ka, sum2)
        for ka in (1, 2, 3, -1, -2):
            w = rnd.getrandbits(18)
            H.hs_sfe_mulw_mag(C.byref(o), C.byref(a), ka, w)
            assert o.value() == ka * a.value() * w % P
        o4 = (Gf * 4)()
        H.hs_sfe_diff_mul(o4, C.byref(a), C.byref(b), C.byref(c))
        d = a.value() - b.value()
        assert o4[0].value() == 2 * c.value() * d % P
        assert o4[1].value() == d * d % P
        assert o4[2].value() == (a.value() + b.value()) ** 2 % P
        assert o4[3].value() == (39081 * d + a.value()) * d % P


@pytest.mark.parametrize("call", ["H.hs_sfe_sqr_mag(o, a, 2, 0)",        # a sum of two products through the plain square
                                  "H.hs_sfe_mul_mag(o, a, a, 2, 2)",      # sum x sum
                                  "H.hs_sfe_sqr_mag(o, a, 3, 1)"])
def test_signed_layer_checker_traps_beyond_the_contract(H, call):
    """The checker is not vacuous: the same entry points abort (SIGILL from __builtin_trap) one step beyond the limits."""
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from _libs import Gf\n"
            "H = C.CDLL(%r); a = Gf()\n"
            "for i in range(8): a.limb[i] = (1 << 56) - 1\n"
            "o = Gf(); a = C.byref(a); o = C.byref(o)\n%s\n") % (os.path.dirname(os.path.abspath(__file__)), H._name, call)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True)
    assert r.returncode < 0, (call, r.returncode, r.stderr[-500:])


def test_scalars(H, O):
    rnd = random.Random(3)
    for it in range(100):
        a, b = Scalar.from_int(rnd.getrandbits(446) % Q), Scalar.from_int(rnd.getrandbits(446) % Q)
        o1, o2 = Scalar(), Scalar()
        O.orc_scalar_mul(C.byref(o1), C.byref(a), C.byref(b)); H.hs_sc_mul(C.byref(o2), C.byref(a), C.byref(b))
        assert bytes(o1) == bytes(o2)
        O.orc_scalar_sub(C.byref(o1), C.byref(a), C.byref(b)); H.hs_sc_sub(C.byref(o2), C.byref(a), C.byref(b))
        assert bytes(o1) == bytes(o2)
        for n in (57, 72, 114):
            x = bytes(rnd.getrandbits(8) for _ in range(n)) if it else b"\xff" * n
            O.orc_scalar_decode_long(C.byref(o1), buf(x), n); H.hs_sc_decode_long(C.byref(o2), buf(x), C.c_size_t(n))
            assert bytes(o1) == bytes(o2)


def test_ladders_and_codecs(H, O):
    rnd = random.Random(4)
    edge = [0, 1, Q - 1, 2**445]
    for it in range(24):
        k = Scalar.from_int(rnd.getrandbits(446) % Q)
        pt = Point(); O.orc_precomputed_scalarmul(C.byref(pt), O.orc_precomputed_base(), C.byref(k))
        s = Scalar.from_int(edge[it] if it < len(edge) else rnd.getrandbits(446) % Q)
        a, b = Point(), Point()
        O.orc_point_scalarmul(C.byref(a), C.byref(pt), C.byref(s)); H.hs_point_scalarmul(C.byref(b), C.byref(pt), C.byref(s))
        assert _enc(O, a) == _enc(O, b) and H.hs_point_valid(C.byref(b)) == -1
        O.orc_precomputed_scalarmul(C.byref(a), O.orc_precomputed_base(), C.byref(s))
        H.hs_precomputed_scalarmul(C.byref(b), O.orc_precomputed_base(), C.byref(s))
        assert _enc(O, a) == _enc(O, b)
        e = (C.c_uint8 * 56)(); H.hs_point_encode(e, C.byref(a)); assert bytes(e) == _enc(O, a)
        d = Point(); assert H.hs_point_decode(C.byref(d), e, 1) == -1 and _enc(O, d) == bytes(e)
        e2, e3 = (C.c_uint8 * 57)(), (C.c_uint8 * 57)()
        O.orc_point_encode_like_eddsa(e2, C.byref(a)); H.hs_point_encode_eddsa(e3, C.byref(a))
        assert bytes(e2) == bytes(e3)
        d2 = Point()
        assert O.orc_point_decode_like_eddsa(C.byref(d2), e2) == H.hs_point_decode_eddsa(C.byref(d), e2)
        t = Scalar.from_int(rnd.getrandbits(446) % Q)
        O.orc_point_double_scalarmul(C.byref(a), C.byref(pt), C.byref(s), C.byref(d2), C.byref(t))
        H.hs_point_double_scalarmul(C.byref(b), C.byref(pt), C.byref(s), C.byref(d2), C.byref(t))
        assert _enc(O, a) == _enc(O, b)


def test_verify(H, O):
    base = O.orc_precomputed_base().contents
    rnd = random.Random(5)
    for it, mlen in enumerate((0, 1, 32, 125, 126, 136, 300)):
        ctx = bytes(rnd.getrandbits(8) for _ in range((0, 3, 255)[it % 3]))
        sigs, pks, msgs = _gen.signatures(O, 2, msglen=mlen, seed=b"hs%d" % it, context=ctx, prehashed=bool(it & 1))
        for j in range(2):
            sig = bytearray(sigs[j].tobytes())
            if j: sig[rnd.randrange(114)] ^= 1 << rnd.randrange(8)
            m = buf(msgs[j]) if msgs[j] else None
            c = buf(ctx) if ctx else None
            want = O.orc_ed448_verify(buf(sig), buf(pks[j].tobytes()), m, mlen, it & 1, c, len(ctx))
            got = H.hs_ed448_verify(buf(sig), buf(pks[j].tobytes()), m, C.c_size_t(mlen), C.c_uint8(it & 1), c,
                                    C.c_uint8(len(ctx)), C.byref(base))
            assert got == want and (j or got == -1)


def test_x448_and_signing(H, O):
    tab = O.orc_precomputed_base()
    rnd = random.Random(6)
    rb = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    for it in range(12):
        b = (bytes(56), b"\xff" * 56, bytes([5] + [0] * 55))[it] if it < 3 else rb(56)
        s = rb(56)
        o1, o2 = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
        assert O.orc_x448(o1, buf(b), buf(s)) == H.hs_x448(o2, buf(b), buf(s)) and bytes(o1) == bytes(o2)
        O.orc_x448_derive_public_key(o1, buf(s)); H.hs_x448_derive_public_key(o2, buf(s), tab)
        assert bytes(o1) == bytes(o2)
        sk, msg, ctx = rb(57), rb((0, 1, 70, 71, 126, 300)[it % 6]), rb((0, 3, 255)[it % 3])
        p1, p2, s1, s2 = (C.c_uint8 * 57)(), (C.c_uint8 * 57)(), (C.c_uint8 * 114)(), (C.c_uint8 * 114)()
        O.orc_ed448_derive_public_key(p1, buf(sk)); H.hs_ed448_derive_public_key(p2, buf(sk), tab)
        assert bytes(p1) == bytes(p2)
        m, c = (buf(msg) if msg else None), (buf(ctx) if ctx else None)
        O.orc_ed448_sign(s1, buf(sk), p1, m, len(msg), it & 1, c, len(ctx))
        H.hs_ed448_sign(s2, buf(sk), p1, m, C.c_size_t(len(msg)), C.c_uint8(it & 1), c, C.c_uint8(len(ctx)), tab)
        assert bytes(s1) == bytes(s2)


@pytest.mark.parametrize("bits", [8, 10, 12, 14, 16, 18, 20, 22, 24])
def test_fixed_base_window_table_ladder(H, O, bits):
    """The base point's window table of signed `bits`-bit digits (S*B of verification, the device's fast path for the
    base point): every width the library can be asked for, its recoding offset and its top digit."""
    H.hs_bwt_scalarmul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    n = 12
    s = _gen.random_scalars(n, b"hs-bwt%d" % bits)
    s[:6] = _gen.scalars_from_ints([0, 1, Q - 1, 2**445, 2, 255])
    out = np.empty((n, 32), np.uint64)
    H.hs_bwt_scalarmul(out.ctypes.data, C.cast(O.orc_precomputed_base(), C.c_void_p), s.ctypes.data, n, bits)
    assert (_gen.oracle_encode(out) == _gen.oracle_encode(_gen.oracle_fixed(O, s))).all()


def test_elligator_and_dual(H, O):
    import json
    H.hs_point_from_hash.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    kats = json.load(open(os.path.join(os.path.dirname(HS_DIR), "golden", "kats.json")))["elligator_nonuniform"]
    for c in kats:
        p = Point()
        H.hs_point_from_hash(C.byref(p), buf(bytes.fromhex(c["hash"])), 0)
        assert _enc(O, p).hex() == c["point"]
    rnd = random.Random(8)
    for it in range(20):
        h = bytes(rnd.getrandbits(8) for _ in range(112)) if it > 1 else (bytes(112), b"\xff" * 112)[it]
        a, b = Point(), Point()
        O.orc_point_from_hash_uniform(C.byref(a), buf(h)); H.hs_point_from_hash(C.byref(b), buf(h), 1)
        assert _enc(O, a) == _enc(O, b)
        s1, s2 = Scalar.from_int(rnd.getrandbits(446) % Q), Scalar.from_int(rnd.getrandbits(446) % Q)
        o1, o2, w1, w2 = Point(), Point(), Point(), Point()
        H.hs_point_dual_scalarmul(C.byref(o1), C.byref(o2), C.byref(a), C.byref(s1), C.byref(s2))
        O.orc_point_scalarmul(C.byref(w1), C.byref(a), C.byref(s1)); O.orc_point_scalarmul(C.byref(w2), C.byref(a), C.byref(s2))
        assert _enc(O, o1) == _enc(O, w1) and _enc(O, o2) == _enc(O, w2)


def test_mac_counts_match_bench(H, O):
    """bench.py prices its kernels in 32x32->64 multiply-accumulates per operation: the figures must be
    what the lane code really executes, counted by the checker accumulator of this host build."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    H.hs_mac_count_of.restype = C.c_ulonglong
    base = np.frombuffer(bytes(O.orc_point_base().contents), np.uint64).copy()
    comb = np.frombuffer(bytes(O.orc_precomputed_base().contents), np.uint64).copy()
    s = _gen.stream_scalars(1, b"mac-count")[0].copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    names = ["fe_mul", "fe_sqr", "fe_mulw", "dbl", "dbl_t", "add_niels_t", "niels_to_pt", "isr", "decode_eddsa",
             "pt_add", "pt_eq", "varbase5", "varbase4", "comb", "comb_big", "ladder", "table16"]
    c = {name: H.hs_mac_count_of(i, p(base), p(s), p(comb)) for i, name in enumerate(names)}
    assert (c["fe_mul"], c["fe_sqr"], c["fe_mulw"]) == (192, 136, 16)
    assert c["dbl"] == 4 * 136 + 3 * 192 and c["dbl_t"] == c["dbl"] + 192
    W = bench.WORKLOADS
    assert W["varbase"]["macs"] == c["varbase5"] == 2175 * 192 + 1785 * 136 + 17 * 16
    # the library's default: the table-free Montgomery ladder (montgomery.hpp) -- count 15 holds its own inversion,
    # the device shares one between the 8 operations a lane owns at batch 2^20 (chain: 3 more multiplications)
    inv = c["isr"] + 2 * c["fe_sqr"] + c["fe_mul"]
    assert (W["varbase"]["macs_ladder"], W["varbase"]["macs_inversion"]) == (c["ladder"] - inv, inv)
    assert W["varbase"]["macs_index_independent"] == c["ladder"] - inv + 3 * c["fe_mul"] + inv // 8
    assert c["ladder"] - inv == 446 * (5 * 192 + 4 * 136 + 16) + (16 * 192 + 2 * 136 + 5 * 16)   # steps + u(P), recovery and the map back
    assert W["fixed"]["macs_reference_comb"] == c["comb"] == c["niels_to_pt"] + 89 * c["add_niels_t"] - 17 * 192 + 17 * c["dbl_t"]
    assert W["fixed"]["macs"] == c["comb_big"]          # large batches: the caller's table re-combed to 4 x 7 x 16
    # the built-in base point with index-independent access: the library's 4 x 7 x 16 comb
    assert W["base"]["macs_index_independent"] == c["comb_big"] == c["niels_to_pt"] + 63 * c["add_niels_t"] - 15 * 192 + 15 * c["dbl_t"]
    # base-point window table, 16-bit digits: one conversion + 27 mixed additions
    assert W["base"]["macs"] == c["niels_to_pt"] + 27 * c["add_niels_t"]
    # verification with half-size scalars (ed448_verify_lattice): two decodings, two window tables, a 45-window
    # ladder over both points, two correcting additions; the base-point half is counted with the comb here and
    # swapped for the 28 window-table additions the device kernel does
    sigs, pks, msgs = _gen.signatures(O, 3, msglen=32, seed=b"mac-count-sig", nkeys=3)
    H.hs_mac_counter_get.restype = C.c_ulonglong
    H.hs_ed448_verify_lattice.restype = C.c_int
    per = []
    for i in range(3):
        m = (C.c_uint8 * 32).from_buffer_copy(msgs[i])
        H.hs_mac_counter_reset()
        assert H.hs_ed448_verify_lattice(p(sigs[i]), p(pks[i]), m, C.c_size_t(32), C.c_uint8(0), None, C.c_uint8(0), p(comb)) == -1
        per.append(H.hs_mac_counter_get())
    assert len(set(per)) == 1                                                                  # the same work for every signature
    own = per[0] - (c["comb"] + c["pt_add"]) + 28 * c["add_niels_t"]
    assert W["verify"]["macs_own_key"] == W["verify_distinct"]["macs"] == own
    assert own < 0.80 * 766_184        # the full-length ladder with one exponentiation (round 2's alternative)
    # a key with a pooled table (shared by the signatures of a batch): its decoding and its table are not the lane's,
    # but one lane's per distinct key: (decoding + table) * 2^10 keys / 2^20 signatures more per signature
    assert W["verify"]["macs_shared_keys"] == own - c["decode_eddsa"] - c["table16"]
    assert W["verify"]["macs_pooled_tables"] == W["verify"]["macs_shared_keys"] + (c["decode_eddsa"] + c["table16"]) * W["verify"]["keys"] // 2**20
    # a key with a comb of its own (keys that sign many of a batch's signatures: BASELINE config 4): R's decoding, the
    # comb ladder, the base point's additions, the comparison; per key: its decoding, 432 doublings, 256 entries of
    # 6 additions each and their normalisation (counted with ONE inversion per key; the device shares one between the
    # keys a lane serves)
    counts = {}
    for wide, hook in ((False, H.hs_ed448_verify_keycomb), (True, H.hs_ed448_verify_keycomb_wide), ("x", H.hs_ed448_verify_keycomb_xwide)):
        hook.restype = C.c_int
        seen = set()
        for i in range(3):
            m = (C.c_uint8 * 32).from_buffer_copy(msgs[i])
            two = []
            for _ in range(2):                                    # the first call with a key builds its comb
                H.hs_mac_counter_reset()
                assert hook(p(sigs[i]), p(pks[i]), m, C.c_size_t(32), C.c_uint8(0), None, C.c_uint8(0), p(comb)) == -1
                two.append(H.hs_mac_counter_get())
            seen.add((two[0] - two[1], two[1]))
        assert len(seen) == 1
        counts[wide] = seen.pop()
    # (the host hook inverts K per signature; the device shares one inversion between the 8 signatures a lane owns
    # at batch 2^20: 3 more multiplications for the chain)
    adjust = 28 * c["add_niels_t"] - (c["comb"] + c["pt_add"]) - inv + 3 * c["fe_mul"] + inv // 8
    per_key, per_sig = counts[False][0], counts[False][1] + adjust
    per_key_wide, per_sig_wide = counts[True][0], counts[True][1] + adjust
    rest = 28 * c["add_niels_t"] + 12 * c["fe_mul"] + 3 * c["fe_sqr"] + c["fe_mulw"] + inv // 8
    assert per_sig == c["comb_big"] + rest
    # the wider comb of keys with hundreds of signatures (4 x 8 x 14): 13 doublings + 55 additions instead of 15 + 63
    assert per_sig_wide == c["niels_to_pt"] + 55 * c["add_niels_t"] - 13 * 192 + 13 * c["dbl_t"] + rest
    assert (W["verify"]["macs_key_comb"], W["verify"]["macs_per_key_comb"]) == (per_sig, per_key)
    assert (W["verify"]["macs_key_comb_wide"], W["verify"]["macs_per_key_comb_wide"]) == (per_sig_wide, per_key_wide)
    # ... and the widest, of keys with a thousand (5 x 9 x 10): 9 doublings + 49 additions
    per_key_xwide, per_sig_xwide = counts["x"][0], counts["x"][1] + adjust
    assert per_sig_xwide == c["niels_to_pt"] + 49 * c["add_niels_t"] - 9 * 192 + 9 * c["dbl_t"] + rest
    assert (W["verify"]["macs_key_comb_xwide"], W["verify"]["macs_per_key_comb_xwide"]) == (per_sig_xwide, per_key_xwide), (per_sig_xwide, per_key_xwide)
    assert W["verify"]["macs"] == per_sig_xwide + per_key_xwide * W["verify"]["keys"] // 2**20    # 2^10 keys x 2^10 signatures: the widest
    assert per_sig_xwide < per_sig_wide < per_sig < 0.30 * W["verify"]["macs_shared_keys"]


def test_big_comb_of_the_base_point_matches_oracle(H, O):
    """The 4 x 7 x 16 comb the index-independent base-point kernels walk (scalarmul.hpp comb_big; the table built
    as k_build_comb_big builds it, entry by entry as scalar multiples of B): s * B against the oracle for edge
    and random scalars."""
    rnd = random.Random(31)
    comb = np.frombuffer(bytes(O.orc_precomputed_base().contents), np.uint64).copy()
    vals = [0, 1, 2, 3, Q - 1, Q - 2, 2**445, 2**444 - 1, 2**16, 2**16 - 1, 2**112, (Q - 1) // 2] + [rnd.getrandbits(446) % Q for _ in range(40)]
    scal = _gen.scalars_from_ints(vals)
    want = _gen.oracle_encode(_gen.oracle_fixed(O, scal))
    for i in range(len(vals)):
        out = np.zeros(32, np.uint64)
        H.hs_comb_big_scalarmul(out.ctypes.data_as(C.c_void_p), comb.ctypes.data_as(C.c_void_p), scal[i].ctypes.data_as(C.c_void_p))
        assert (_gen.oracle_encode(out.reshape(1, 32))[0] == want[i]).all(), hex(vals[i])


@pytest.mark.parametrize("rcp_error", [0.0, 2.0 ** -22, -2.0 ** -22, 2.0 ** -12])
def test_half_size_pair_of_a_challenge(H, rcp_error):
    """lattice.hpp: (rho, tau) with rho == tau * h (mod q), 0 <= rho < 2^223, 0 < |tau| < 2^223: exactly the
    first pair below 2^223 of the remainder sequence of (q, h), for random and degenerate challenges (0, 1,
    small, q - 1, around 2^223, huge first quotients).
    rcp_error: the device estimates its single-precision quotients with v_rcp_f64 (relative error about 2^-23),
    the host build with an exact division; the host's reciprocal is perturbed by that much either way (and by a
    gross 2^-12) to run the paths the device runs -- the estimate's accuracy may cost steps, never the result."""
    H.hs_set_rcp_perturb.argtypes = [C.c_double]
    H.hs_set_rcp_perturb(rcp_error)
    try:
        _half_size_pairs(H, 3000 if rcp_error == 0.0 else 600)
    finally:
        H.hs_set_rcp_perturb(0.0)


def _half_size_pairs(H, nrandom):
    rnd = random.Random(23)
    cases = [0, 1, 2, 3, Q - 1, Q - 2, 2**223, 2**223 - 1, 2**223 + 1, 2**224, 2**224 + 1, (Q - 1) // 2, (Q + 1) // 2,
             2**445, 2**300 + 1] + [(Q // k) % Q for k in (3, 5, 7, 2**30 + 1, 2**31 - 1, 2**62 + 1, 2**100 + 7, 2**222 + 1)]
    cases += [rnd.getrandbits(446) % Q for _ in range(nrandom)] + [rnd.getrandbits(b) for b in (10, 100, 222, 223, 224, 225, 300)]
    for h in cases:
        rho = (C.c_uint32 * 15)(); tau = (C.c_uint32 * 8)()
        H.hs_half_size_pair(rho, tau, C.byref(Scalar.from_int(h)))
        r = sum(int(rho[i]) << (32 * i) for i in range(15))
        t = sum(int(tau[i]) << (32 * i) for i in range(8))
        if t >> 255:
            t -= 1 << 256
        r0, r1, t0, t1 = Q, h, 0, 1                          # the remainder sequence in Python integers
        while r1 >= 2**223:
            k = r0 // r1
            r0, r1, t0, t1 = r1, r0 - k * r1, t1, t0 - k * t1
        assert (r, t) == (r1, t1), hex(h)                    # Lehmer's single-precision quotients are the exact ones
        assert t != 0 and abs(t) < 2**223 and 0 <= r < 2**223 and (r - t * h) % Q == 0


def test_verification_with_half_size_scalars(H, O):
    """ed448_verify_lattice (eddsa.hpp): the verdicts of the oracle -- which are the reference's -- on valid,
    corrupted and degenerate signatures, the torsion-malleable cases of fixture F7 included (the scalars act
    modulo q in the subgroup of prime order every decoded point lies in, so the accept set is the reference's)."""
    import json
    tab = O.orc_precomputed_base()
    n = 40
    sigs, pks, msgs = _gen.signatures(O, n, msglen=32, seed=b"lattice", nkeys=5)
    msgs = np.frombuffer(b"".join(msgs), np.uint8).reshape(n, 32).copy()
    P_ = 2**448 - 2**224 - 1
    enc = lambda y, s=0: np.frombuffer(int(y).to_bytes(56, "little") + bytes([0x80 * s]), np.uint8)
    special_r = [enc(0), enc(0, 1), enc(1), enc(P_ - 1), enc(P_), enc(2**448 - 1), enc(1, 1), enc(5), enc(5, 1)]
    for i, r in enumerate(special_r):
        sigs[2 + 3 * i, :57] = r
    sigs[30, 56] ^= 0x80
    sigs[31, 56] |= 0x01
    sigs[32, 60] ^= 1
    pks[33] = enc(1)
    pks[34] = enc(P_ - 1)
    pks[35] = enc(0)
    pks[36, 56] ^= 0x80
    msgs[37, 3] ^= 1
    sigs[38, 57:] = 0xff                                             # S >= q: reduced, not rejected (src/eddsa.c:316-327)
    mlist = [m.tobytes() for m in msgs]
    f7 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f7_verify_torsion.json")))["cases"]
    H.hs_ed448_verify_lattice.restype = C.c_int
    H.hs_ed448_verify_lattice_shared_key.restype = C.c_int
    H.hs_ed448_verify_keycomb.restype = C.c_int
    H.hs_ed448_verify_keycomb_wide.restype = C.c_int
    H.hs_ed448_verify_keycomb_xwide.restype = C.c_int
    want = _gen.oracle_verify(O, sigs, pks, mlist)
    for i in range(n):
        m = (C.c_uint8 * len(mlist[i])).from_buffer_copy(mlist[i])
        got = H.hs_ed448_verify_lattice(sigs[i].ctypes.data_as(C.c_void_p), pks[i].ctypes.data_as(C.c_void_p), m,
                                        C.c_size_t(len(mlist[i])), C.c_uint8(0), None, C.c_uint8(0), tab)
        assert got == want[i], i
        # ... and with the key decoded and its table built beforehand (a key shared by the signatures of a batch:
        # one table of +A serves both signs of tau)
        got = H.hs_ed448_verify_lattice_shared_key(sigs[i].ctypes.data_as(C.c_void_p), pks[i].ctypes.data_as(C.c_void_p), m,
                                                   C.c_size_t(len(mlist[i])), C.c_uint8(0), None, C.c_uint8(0), tab)
        assert got == want[i], ("shared key", i)
        # ... and with a fixed-base comb of the key (a key that signs many of a batch's signatures): the reference's
        # equation without a ladder
        got = H.hs_ed448_verify_keycomb(sigs[i].ctypes.data_as(C.c_void_p), pks[i].ctypes.data_as(C.c_void_p), m,
                                        C.c_size_t(len(mlist[i])), C.c_uint8(0), None, C.c_uint8(0), tab)
        assert got == want[i], ("key comb", i)
        got = H.hs_ed448_verify_keycomb_wide(sigs[i].ctypes.data_as(C.c_void_p), pks[i].ctypes.data_as(C.c_void_p), m,
                                             C.c_size_t(len(mlist[i])), C.c_uint8(0), None, C.c_uint8(0), tab)
        assert got == want[i], ("wide key comb", i)
        got = H.hs_ed448_verify_keycomb_xwide(sigs[i].ctypes.data_as(C.c_void_p), pks[i].ctypes.data_as(C.c_void_p), m,
                                              C.c_size_t(len(mlist[i])), C.c_uint8(0), None, C.c_uint8(0), tab)
        assert got == want[i], ("widest key comb", i)
    assert (want == -1).sum() >= 10 and (want == 0).sum() >= 10
    accepted = rejected = 0
    for c in f7:
        sig, pk, msg, ctx = (bytes.fromhex(c[k]) for k in ("sig", "pk", "msg", "ctx"))
        mb = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg or b"\0")
        cb = (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0")
        got = H.hs_ed448_verify_lattice(sig, pk, mb, C.c_size_t(len(msg)), C.c_uint8(c["prehashed"]), cb, C.c_uint8(len(ctx)), tab)
        assert got == c["verdict"], c["kind"]
        assert H.hs_ed448_verify_lattice_shared_key(sig, pk, mb, C.c_size_t(len(msg)), C.c_uint8(c["prehashed"]), cb,
                                                    C.c_uint8(len(ctx)), tab) == c["verdict"], ("shared key", c["kind"])
        assert H.hs_ed448_verify_keycomb(sig, pk, mb, C.c_size_t(len(msg)), C.c_uint8(c["prehashed"]), cb,
                                         C.c_uint8(len(ctx)), tab) == c["verdict"], ("key comb", c["kind"])
        assert H.hs_ed448_verify_keycomb_wide(sig, pk, mb, C.c_size_t(len(msg)), C.c_uint8(c["prehashed"]), cb,
                                              C.c_uint8(len(ctx)), tab) == c["verdict"], ("wide key comb", c["kind"])
        assert H.hs_ed448_verify_keycomb_xwide(sig, pk, mb, C.c_size_t(len(msg)), C.c_uint8(c["prehashed"]), cb,
                                               C.c_uint8(len(ctx)), tab) == c["verdict"], ("widest key comb", c["kind"])
        accepted += got == -1; rejected += got == 0
    assert accepted >= 4 and rejected >= 4


def test_table_free_ladder_matches_oracle_and_golden_f1(H, O):
    """montgomery.hpp: the index-independent variable-base multiplication (Montgomery ladder on the
    Montgomery model of the reference's curve + Okeya-Sakurai recovery) against the oracle's windowed
    multiplication (src/goldilocks.c:405-465): edge scalars (0, 1, q-1 -- the cases where the recovery
    degenerates and a select takes over), the identity and the 2-torsion point as bases, rescaled
    representatives, scalars >= q, and every 8th vector of the reference's fixture F1."""
    def run(bases, scal):
        out = np.empty((len(scal), 32), dtype=np.uint64)
        for i in range(len(scal)):
            H.hs_point_scalarmul_ladder(out[i].ctypes.data_as(C.c_void_p), bases[i].ctypes.data_as(C.c_void_p),
                                        scal[i].ctypes.data_as(C.c_void_p))
        return out
    vals = [0, 1, 2, 3, Q - 1, Q - 2, Q - 3, (Q + 1) // 2, (Q - 1) // 2, 2**445, 2**445 - 1, 2**446 - 1 - Q, 7, 8]
    bases = _gen.oracle_fixed(O, _gen.stream_scalars(len(vals), b"ml/base"))
    scal = _gen.scalars_from_ints(vals)
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases, scal))
    got = run(bases, scal)
    assert (_gen.oracle_encode(got) == want).all()
    for i in range(len(vals)):   # complete extended points: on the curve, X Y = Z T
        assert H.hs_point_valid(got[i].ctypes.data_as(C.c_void_p)) == -1, hex(vals[i])
    # bases with a 2-torsion component, P + (0, -1) = (-x, -y): order 2q, the same class as s*P for every s
    shifted = bases.copy()
    for i in range(len(vals)):
        for fld in (0, 8):                               # X and Y negated limb-wise mod p (T = XY/Z unchanged)
            v = sum(int(shifted[i][fld + k]) << (56 * k) for k in range(8)) % P
            shifted[i][fld:fld + 8] = np.frombuffer(Gf.from_int((P - v) % P), np.uint64)
    assert (_gen.oracle_encode(_gen.oracle_varbase(O, shifted, scal)) == want).all()      # the reference agrees
    assert (_gen.oracle_encode(run(shifted, scal)) == want).all()
    # the identity and (0, -1) as bases: the identity's class whatever the scalar
    ident = np.zeros(32, np.uint64); ident[8] = 1; ident[16] = 1
    t2 = ident.copy(); t2[8:16] = np.frombuffer(Gf.from_int(P - 1), np.uint64)
    special = np.stack([ident, t2, ident, t2])
    s4 = _gen.scalars_from_ints([0, 5, Q - 1, Q - 1])
    enc = _gen.oracle_encode(run(special, s4))
    assert (enc == 0).all()
    # a scalar that is not reduced (q + 5, 2q + 1 as raw words): the reference reduces by its recoding
    raw = np.empty((2, 7), np.uint64)
    raw[0] = np.frombuffer((Q + 5).to_bytes(56, "little"), np.uint64)
    raw[1] = np.frombuffer((2 * Q + 1).to_bytes(56, "little"), np.uint64)
    b2 = bases[:2]
    assert (_gen.oracle_encode(run(b2, raw)) == _gen.oracle_encode(_gen.oracle_varbase(O, b2, raw))).all()
    # golden F1 (the reference's own outputs)
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f1_varbase.npz"))
    idx = list(range(0, 1024, 8)) + list(range(1016, 1024))
    fb = np.empty((len(idx), 32), np.uint64)
    for j, i in enumerate(idx):
        p = Point()
        assert O.orc_point_decode(C.byref(p), buf(d["base"][i].tobytes()), 1) == -1
        fb[j] = np.frombuffer(bytes(p), np.uint64)
    assert (_gen.oracle_encode(run(fb, d["scalar"][idx])) == d["out"][idx]).all()


def test_direct_scalarmul_through_the_ladder_and_its_base_point_fallback(H, O):
    """goldilocks_448_direct_scalarmul (src/goldilocks.c:888-903) the way k_direct_scalarmul_ct computes it: the decoder
    that also yields u(P), the table-free ladder, the encoder -- and, when the encoding does not decode and the call
    does not short-circuit, the BASE POINT multiplied instead, from its constant u(B) (tools/gen_tables.py)."""
    H.hs_direct_scalarmul_ladder.restype = C.c_int
    n = 24
    s = _gen.random_scalars(n, b"hs-direct-s")
    base = _gen.oracle_encode(_gen.oracle_fixed(O, _gen.random_scalars(n, b"hs-direct-b")))
    base[5] = 0                       # identity encoding
    base[6] = 0xff                    # not a field element
    base[7, 0] |= 1                   # negative s
    for allow_id in (0, 1):
        for i in range(n):
            want, got = (C.c_uint8 * 56)(), (C.c_uint8 * 56)()
            r = O.orc_direct_scalarmul(want, base[i].ctypes.data, C.cast(s[i].ctypes.data, C.POINTER(Scalar)), allow_id, 0)
            r2 = H.hs_direct_scalarmul_ladder(got, base[i].ctypes.data_as(C.c_void_p), s[i].ctypes.data_as(C.c_void_p), allow_id)
            assert r == r2 and bytes(want) == bytes(got), (i, allow_id, r, r2)


def test_two_dimensional_ladder_matches_oracle(H, O):
    """montgomery2d.hpp: s1*P1 + s2*P2 on ONE chain of a doubling and two differential additions per bit (the table-free
    counterpart of the reference's shared doubling chain, src/goldilocks.c:467-541), on the checker build: random points
    and scalars; every pair of the scalars 0, 1, 2, q-1, 2^445 and the all-ones pattern; scalars >= q; and the exceptional
    inputs its substitution exists for -- the identity and the 2-torsion point as either base, P2 = +-P1 (also with a
    2-torsion component), either base equal to +-B or +-2B beside a trivial partner -- each of which must be seen to take
    the substitution's path and still produce the oracle's group element."""
    from _libs import Point
    H.hs_double_scalarmul_2d.restype = C.c_int
    g = np.frombuffer(bytes(O.orc_point_base().contents), np.uint64).copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(b1, s1, b2, s2):
        out, what = np.empty((len(s1), 32), dtype=np.uint64), []
        for i in range(len(s1)):
            what.append(H.hs_double_scalarmul_2d(p(out[i]), p(b1[i]), p(s1[i]), p(b2[i]), p(s2[i]), p(g)))
            assert H.hs_point_valid(p(out[i])) == -1, i
        return out, what

    def neg(pt):      # -P: X and T negated limb-wise mod p
        o = pt.copy()
        for fld in (0, 24):
            v = sum(int(o[fld + k]) << (56 * k) for k in range(8)) % P
            o[fld:fld + 8] = np.frombuffer(Gf.from_int((P - v) % P), np.uint64)
        return o

    def shift(pt):    # P + (0, -1) = (-x, -y): the same class
        o = pt.copy()
        for fld in (0, 8):
            v = sum(int(o[fld + k]) << (56 * k) for k in range(8)) % P
            o[fld:fld + 8] = np.frombuffer(Gf.from_int((P - v) % P), np.uint64)
        return o

    n = 48
    b1 = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"ml2/b1"))
    b2 = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"ml2/b2"))
    s1, s2 = _gen.stream_scalars(n, b"ml2/s1"), _gen.stream_scalars(n, b"ml2/s2")
    got, what = run(b1, s1, b2, s2)
    assert (_gen.oracle_encode(got) == _gen.oracle_encode(_gen.oracle_double(O, b1, s1, b2, s2))).all() and not any(what)
    # every pair of edge scalars on one pair of points
    vals = [0, 1, 2, Q - 1, Q - 2, 2**445, 2**446 - 1 - Q, (Q - 1) // 2, int("01" * 223, 2), int("0011" * 111, 2)]
    pairs = [(x, y) for x in vals for y in vals]
    es1, es2 = _gen.scalars_from_ints([x for x, _ in pairs]), _gen.scalars_from_ints([y for _, y in pairs])
    eb1, eb2 = np.repeat(b1[:1], len(pairs), axis=0), np.repeat(b2[:1], len(pairs), axis=0)
    got, what = run(eb1, es1, eb2, es2)
    assert (_gen.oracle_encode(got) == _gen.oracle_encode(_gen.oracle_double(O, eb1, es1, eb2, es2))).all() and not any(what)
    # scalars that are not reduced (the entry point takes any 448-bit value)
    raw = np.frombuffer(_gen.stream(b"ml2/raw", 56 * 8), np.uint64).reshape(8, 7).copy()
    got, _ = run(b1[:8], raw, b2[:8], raw[::-1].copy())
    assert (_gen.oracle_encode(got) == _gen.oracle_encode(_gen.oracle_double(O, b1[:8], raw, b2[:8], raw[::-1].copy()))).all()
    # exceptional inputs: (P1, P2, what ml2_effective must report)
    ident = np.zeros(32, np.uint64); ident[8] = 1; ident[16] = 1
    t2 = ident.copy(); t2[8:16] = np.frombuffer(Gf.from_int(P - 1), np.uint64)
    two_g = np.empty(32, np.uint64)
    O.orc_point_double(C.cast(p(two_g), C.POINTER(Point)), C.cast(p(g), C.POINTER(Point)))
    A, B_ = b1[3], b2[3]
    cases = [(ident, B_, 1 | 4), (t2, B_, 1 | 4), (A, ident, 2 | 8), (A, t2, 2 | 8), (ident, t2, 1 | 2 | 4 | 8),
             (A, A, 1 | 2 | 8), (A, neg(A), 1 | 2 | 8), (A, shift(A), 1 | 2 | 8), (shift(A), neg(A), 1 | 2 | 8),
             (ident, g, 1 | 4), (ident, neg(g), 1 | 4), (ident, two_g, 1 | 4), (g, ident, 2 | 8), (two_g, t2, 2 | 8),
             (g, g, 1 | 2 | 8), (two_g, neg(two_g), 1 | 2 | 8), (g, two_g, 0), (g, A, 0)]
    cb1 = np.array([c[0] for c in cases]); cb2 = np.array([c[1] for c in cases])
    cs1, cs2 = s1[:len(cases)].copy(), s2[:len(cases)].copy()
    got, what = run(cb1, cs1, cb2, cs2)
    assert (_gen.oracle_encode(got) == _gen.oracle_encode(_gen.oracle_double(O, cb1, cs1, cb2, cs2))).all()
    for i, c in enumerate(cases):
        assert what[i] & c[2] == c[2] and (c[2] != 0 or what[i] == 0), (i, what[i], c[2])
    # ... and with the scalars that make the substituted sums degenerate: s1 + s2 = 0, s1 - s2 = 0
    z1 = _gen.scalars_from_ints([5, 5, Q - 5, 7])
    z2 = _gen.scalars_from_ints([Q - 5, 5, Q - 5, 0])
    zb1 = np.array([A, A, A, A]); zb2 = np.array([A, neg(A), A, A])
    got, _ = run(zb1, z1, zb2, z2)
    assert (_gen.oracle_encode(got) == _gen.oracle_encode(_gen.oracle_double(O, zb1, z1, zb2, z2))).all()
