"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: bit-exact on canonical encodings / status words (integer arithmetic)."""
import ctypes as C

import numpy as np
import pytest

import _gen
from _libs import Q, P

pytestmark = pytest.mark.gpu

EDGE_SCALARS = [0, 1, 2, Q - 1, Q - 2, 2**445, 2**445 - 1, (Q + 1) // 2, 31, 32, 2**224, 2**440 + 12345]


@pytest.fixture(autouse=True, params=["index_independent", "fast", "lane_kernels_only"])
def table_mode(request, ga):
    """Every test of this module runs under both table-access policies (include/goldilocks_amd.h): the
    library's default (index-independent scans / LDS comb) and the opt-in digit-addressed tables -- and a
    third time with the one-operation-per-wave path for small batches switched off, so that small inputs
    reach the lane-per-operation kernels too."""
    default = ga.get_wave_batch_max()
    ga.set_table_access(ga.TABLES_FAST if request.param == "fast" else ga.TABLES_INDEX_INDEPENDENT)
    ga.set_wave_batch_max(0 if request.param == "lane_kernels_only" else default)
    yield request.param
    ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
    ga.set_wave_batch_max(default)


def enc(ga, pts):
    """canonical bytes of the GPU's raw points -- by the GPU's own encoder AND by the oracle's (orc_point_encode on the
    raw limbs): a comparison with the oracle's results never rests on the device encoder alone"""
    got = ga.point_encode_batch(pts)
    assert (got == _gen.oracle_encode(pts)).all(), "k_point_encode and the oracle's encoder disagree on the GPU's raw points"
    return got


def test_device_is_gfx950(ga):
    info = ga.device_info()
    assert info["arch"].startswith("gfx950") and info["compute_units"] > 0


def test_field_ops_vs_oracle(ga, O):
    import torch
    n = 4096
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    b = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    # weakly reduced / unreduced-limb inputs and special values
    a[0] = 0; b[0] = 0
    a[1] = 2**56 - 1; b[1] = 2**56 - 1
    a[2] = 2**56 + 255; b[2] = 2**56 + 255
    a[3] = [1, 0, 0, 0, 0, 0, 0, 0]
    a[4] = [2**56 - 1] * 4 + [2**56 - 2] + [2**56 - 1] * 3          # p itself
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.empty_like(da)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    from _libs import Gf
    def ser(limbs):
        g = Gf(); g.limb[:] = [int(x) for x in limbs]
        buf = (C.c_uint8 * 56)(); O.orc_gf_serialize(buf, C.byref(g)); return bytes(buf)
    for op in (0, 1, 2, 3):
        ga.dev("field_op", out.data_ptr(), st.data_ptr(), da.data_ptr(), db.data_ptr(), op, n, None)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        status = st.cpu().numpy()
        for i in list(range(8)) + list(range(8, n, 37)):
            ga_, gb_, go = Gf(), Gf(), Gf()
            ga_.limb[:] = [int(x) for x in a[i]]; gb_.limb[:] = [int(x) for x in b[i]]
            if op == 0: O.orc_gf_mul(C.byref(go), C.byref(ga_), C.byref(gb_))
            elif op == 1: O.orc_gf_sqr(C.byref(go), C.byref(ga_))
            elif op == 2:
                m = O.orc_gf_isr(C.byref(go), C.byref(ga_))
                assert (m != 0) == (status[i] != 0), (op, i)
            else:
                go = ga_; O.orc_gf_strong_reduce(C.byref(go))
                assert [int(x) for x in got[i]] == list(go.limb), ("strong", i)
            assert ser(got[i]) == ser(go.limb), (op, i)


def test_fixed_base_vs_oracle(ga, O):
    n = 2048
    s = _gen.random_scalars(n, b"t-fixed")
    s[:len(EDGE_SCALARS)] = _gen.scalars_from_ints(EDGE_SCALARS)
    got = enc(ga, ga.precomputed_scalarmul_batch(s))
    want = _gen.oracle_encode(_gen.oracle_fixed(O, s))
    assert (got == want).all()


def test_variable_base_vs_oracle(ga, O):
    n = 1024
    s = _gen.random_scalars(n, b"t-var-s")
    bases = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-var-b"))
    s[:len(EDGE_SCALARS)] = _gen.scalars_from_ints(EDGE_SCALARS)
    bases[20] = ga.point_identity()
    bases[21] = ga.point_base()
    got = ga.point_scalarmul_batch(bases, s)
    want = _gen.oracle_varbase(O, bases, s)
    assert (enc(ga, got) == _gen.oracle_encode(want)).all()
    # outputs must be complete extended points (XY = ZT, on curve)
    import torch
    d = torch.from_numpy(got.view(np.int64)).cuda()
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", st.data_ptr(), d.data_ptr(), None, 1, n, None)
    assert (st.cpu().numpy() == -1).all()


def test_ragged_and_empty_batches(ga, O):
    assert ga.point_scalarmul_batch(np.empty((0, 32), np.uint64), np.empty((0, 7), np.uint64)).shape == (0, 32)
    for n in (1, 63, 65, 257, 1000):
        s = _gen.random_scalars(n, b"t-ragged%d" % n)
        bases = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-ragged-b%d" % n))
        got = enc(ga, ga.point_scalarmul_batch(bases, s))
        assert (got == _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))).all(), n


def test_output_may_alias_input_single_op(ga, O):
    s = _gen.random_scalars(1, b"t-alias")
    base = _gen.oracle_fixed(O, _gen.random_scalars(1, b"t-alias-b"))
    buf = base.copy()
    ga.lib().goldilocks_448_point_scalarmul(buf.ctypes.data, buf.ctypes.data, s.ctypes.data)
    assert (enc(ga, buf) == _gen.oracle_encode(_gen.oracle_varbase(O, base, s))).all()


def test_double_scalarmul_vs_oracle(ga, O):
    from _libs import Point, Scalar
    n = 256
    s1, s2 = _gen.random_scalars(n, b"t-d1"), _gen.random_scalars(n, b"t-d2")
    b1 = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-db1"))
    b2 = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-db2"))
    got = enc(ga, ga.point_double_scalarmul_batch(b1, s1, b2, s2))
    gotb = enc(ga, ga.point_double_scalarmul_batch(None, s1, b2, s2))
    want = np.empty((n, 32), np.uint64); wantb = np.empty((n, 32), np.uint64)
    for i in range(n):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        O.orc_point_double_scalarmul(C.cast(p(want[i]), C.POINTER(Point)), C.cast(p(b1[i]), C.POINTER(Point)),
                                     C.cast(p(s1[i]), C.POINTER(Scalar)), C.cast(p(b2[i]), C.POINTER(Point)),
                                     C.cast(p(s2[i]), C.POINTER(Scalar)))
        O.orc_base_double_scalarmul_non_secret(C.cast(p(wantb[i]), C.POINTER(Point)),
                                               C.cast(p(s1[i]), C.POINTER(Scalar)),
                                               C.cast(p(b2[i]), C.POINTER(Point)),
                                               C.cast(p(s2[i]), C.POINTER(Scalar)))
    assert (got == _gen.oracle_encode(want)).all()
    assert (gotb == _gen.oracle_encode(wantb)).all()


def test_decode_encode_and_rejects(ga, O):
    from _libs import Point
    n = 512
    pts = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-enc"))
    ser = enc(ga, pts)
    assert (ser == _gen.oracle_encode(pts)).all()
    dec, st = ga.point_decode_batch(ser)
    assert (st == -1).all() and (enc(ga, dec) == ser).all()
    # random strings: accept/reject must match the oracle
    rnd = np.frombuffer(_gen.stream(b"t-dec-rnd", 56 * n), dtype=np.uint8).reshape(n, 56).copy()
    rnd[0] = 0                       # identity encoding
    rnd[1] = 0xff                    # >= p
    rnd[2] = ser[2]; rnd[2, 0] |= 1  # "negative" s
    special = [1, 2, 3, 4, 5, P - 1, P - 2, P, P + 1, 2**447, (P - 1) // 2, (P + 1) // 2]
    for row, v in enumerate(special, start=3):                       # hand-picked encodings
        rnd[row] = np.frombuffer(v.to_bytes(56, "little"), np.uint8)
    _, st0 = ga.point_decode_batch(rnd, allow_identity=False)
    _, st1 = ga.point_decode_batch(rnd, allow_identity=True)
    for i in range(n):
        p = Point()
        buf = (C.c_uint8 * 56).from_buffer_copy(rnd[i].tobytes())
        assert O.orc_point_decode(C.byref(p), buf, 0) == st0[i], i
        assert O.orc_point_decode(C.byref(p), buf, 1) == st1[i], i
    assert st0[0] == 0 and st1[0] == -1 and st0[1] == 0


def test_eddsa_encode_decode(ga, O):
    from _libs import Point
    n = 256
    pts = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-eddsa"))
    e = ga.point_encode_like_eddsa_batch(pts)
    for i in range(0, n, 5):
        buf = (C.c_uint8 * 57)()
        O.orc_point_encode_like_eddsa(buf, C.cast(pts[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)))
        assert bytes(buf) == e[i].tobytes()
    dec, st = ga.point_decode_like_eddsa_batch(e)
    assert (st == -1).all()
    # decode(encode(P)) = 4P
    four = _gen.oracle_varbase(O, pts, _gen.scalars_from_ints([4] * n))
    assert (enc(ga, dec) == _gen.oracle_encode(four)).all()
    bad = e.copy(); bad[:, 56] |= 0x01          # byte 56 must be 0x00 / 0x80
    _, st = ga.point_decode_like_eddsa_batch(bad)
    assert (st == 0).all()
    rnd = np.frombuffer(_gen.stream(b"t-eddsa-rnd", 57 * n), dtype=np.uint8).reshape(n, 57).copy()
    rnd[:, 56] &= 0x80
    special = [0, 1, 2, 3, P - 1, P - 2, P, P + 1, 2**447, 2**448 - 1, (P - 1) // 2, (P + 1) // 2]
    for k, v in enumerate(special):                                  # hand-picked y, both sign bits
        for sgn in (0, 1):
            rnd[2 * k + sgn, :56] = np.frombuffer(v.to_bytes(56, "little"), np.uint8)
            rnd[2 * k + sgn, 56] = 0x80 * sgn
    got, st = ga.point_decode_like_eddsa_batch(rnd)
    for i in range(n):
        p = Point()
        assert O.orc_point_decode_like_eddsa(C.byref(p), (C.c_uint8 * 57).from_buffer_copy(rnd[i].tobytes())) == st[i], i
        if st[i] == -1 and i < 2 * len(special):
            w = np.frombuffer(bytes(p), np.uint64).reshape(1, 32)
            assert (enc(ga, got[i:i + 1]) == _gen.oracle_encode(w)).all(), i


def test_group_ops(ga, O):
    import torch
    n = 256
    a = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-ga"))
    sa, sb = _gen.random_scalars(n, b"t-ga"), _gen.random_scalars(n, b"t-gb")
    b = _gen.oracle_fixed(O, sb)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.empty_like(da)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    tot = lambda f: _gen.scalars_from_ints([f(int.from_bytes(x.tobytes(), "little"), int.from_bytes(y.tobytes(), "little"))
                                            for x, y in zip(sa, sb)])
    for op, f in ((0, lambda x, y: x + y), (1, lambda x, y: x - y), (2, lambda x, y: 2 * x)):
        ga.dev("point_op", out.data_ptr(), da.data_ptr(), db.data_ptr(), op, n, None)
        got = out.cpu().numpy().view(np.uint64)
        assert (enc(ga, got) == _gen.oracle_encode(_gen.oracle_fixed(O, tot(f)))).all(), op
    ga.dev("point_pred", st.data_ptr(), da.data_ptr(), da.data_ptr(), 0, n, None)
    assert (st.cpu().numpy() == -1).all()
    ga.dev("point_pred", st.data_ptr(), da.data_ptr(), db.data_ptr(), 0, n, None)
    assert (st.cpu().numpy() == 0).all()
    junk = a.copy(); junk[:, 3] ^= np.uint64(5)
    dj = torch.from_numpy(junk.view(np.int64)).cuda()
    ga.dev("point_pred", st.data_ptr(), dj.data_ptr(), None, 1, n, None)
    assert (st.cpu().numpy() == 0).all()


def test_scalar_api_vs_oracle(ga, O):
    """The reference's scalar API (point_448.h:100-260, src/scalar.c:30-332) on the device (k_scalar_op): add, sub, mul,
    halve, invert, decode with its range check, decode_long at lengths 0 ... 250 -- every lane against the oracle and
    against Python's integers; edge values 0, 1, q - 1, q, 2^448 - 1; then the drop-in names one call at a time."""
    from _libs import Scalar
    n = 300
    rnd = np.random.default_rng(21)
    ints = [0, 1, 2, Q - 1, Q - 2, (Q + 1) // 2] + [int.from_bytes(rnd.bytes(56), "little") % Q for _ in range(n - 6)]
    a = _gen.scalars_from_ints(ints)
    b = _gen.scalars_from_ints(ints[3:] + ints[:3])
    tob = lambda xs: np.frombuffer(b"".join((x % Q).to_bytes(56, "little") for x in xs), np.uint64).reshape(-1, 7)
    L = ga.lib()
    def run(op, x, y=None, length=0, status=False):
        out = np.zeros((n, 7), np.uint64)
        st = np.full(n, 7, np.int32)
        ga._check(L.goldilocks_amd_scalar_op_batch(out.ctypes.data, st.ctypes.data if status else None, x.ctypes.data,
                                                   y.ctypes.data if y is not None else None, op, length, n))
        return (out, st) if status else out
    ia, ib = ints, ints[3:] + ints[:3]
    assert (run(0, a, b) == tob([x + y for x, y in zip(ia, ib)])).all()
    assert (run(1, a, b) == tob([x - y for x, y in zip(ia, ib)])).all()
    assert (run(2, a, b) == tob([x * y for x, y in zip(ia, ib)])).all()
    assert (ga.scalar_op_batch("mul", a, b) == run(2, a, b)).all() and (ga.scalar_op_batch("halve", a) == run(3, a)).all()   # the package's wrapper
    assert (run(3, a) == tob([x * pow(2, -1, Q) for x in ia])).all()
    inv, st = run(4, a, status=True)
    assert (inv == tob([pow(x, -1, Q) if x else 0 for x in ia])).all() and (st == [-1 if x else 0 for x in ia]).all()
    for i in (0, 1, 3, 17):       # ... and the oracle's own word on a few
        o = Scalar()
        assert O.orc_scalar_invert(C.byref(o), C.cast(a[i].ctypes.data_as(C.c_void_p), C.POINTER(Scalar))) == st[i] and bytes(o) == inv[i].tobytes()
    raws = [0, Q - 1, Q, Q + 1, 2**448 - 1, 2**446] + [int.from_bytes(rnd.bytes(56), "little") for _ in range(n - 6)]
    ser = np.frombuffer(b"".join(x.to_bytes(56, "little") for x in raws), np.uint8).reshape(n, 56).copy()
    dec, st = run(5, ser, status=True)
    assert (dec == tob(raws)).all() and (st == [-1 if x < Q else 0 for x in raws]).all()
    dec2, st2 = ga.scalar_op_batch("decode", ser)
    assert (dec2 == dec).all() and (st2 == st).all()
    for length in (0, 1, 55, 56, 57, 72, 112, 113, 114, 250):
        blob = np.frombuffer(rnd.bytes(max(length * n, 1)), np.uint8).copy()
        want = tob([int.from_bytes(blob[length * i:length * (i + 1)].tobytes(), "little") for i in range(n)])
        assert (run(6, blob, length=length) == want).all(), length
        if length:
            assert (ga.scalar_op_batch("decode_long", blob[:length * n], length=length) == want).all()
        o = Scalar()
        O.orc_scalar_decode_long(C.byref(o), blob[length * 5:].ctypes.data_as(C.c_void_p), length)
        assert bytes(o) == want[5].tobytes()
    # the drop-in names
    x, y, o = a[7].copy(), b[7].copy(), np.zeros(7, np.uint64)
    L.goldilocks_448_scalar_mul(o.ctypes.data, x.ctypes.data, y.ctypes.data); assert (o == tob([ia[7] * ib[7]])[0]).all()
    L.goldilocks_448_scalar_add(x.ctypes.data, x.ctypes.data, y.ctypes.data); assert (x == tob([ia[7] + ib[7]])[0]).all()   # in place
    L.goldilocks_448_scalar_sub(o.ctypes.data, x.ctypes.data, y.ctypes.data); assert (o == a[7]).all()
    L.goldilocks_448_scalar_halve(o.ctypes.data, a[7].ctypes.data); assert (o == tob([ia[7] * pow(2, -1, Q)])[0]).all()
    assert L.goldilocks_448_scalar_invert(o.ctypes.data, a[7].ctypes.data) == -1 and (o == tob([pow(ia[7], -1, Q)])[0]).all()
    assert L.goldilocks_448_scalar_invert(o.ctypes.data, a[0].ctypes.data) == 0 and not o.any()
    assert L.goldilocks_448_scalar_decode(o.ctypes.data, ser[2].ctypes.data) == 0 and not o.any()           # q itself: rejected, reads as 0
    assert L.goldilocks_448_scalar_decode(o.ctypes.data, ser[1].ctypes.data) == -1 and (o == tob([Q - 1])[0]).all()
    L.goldilocks_448_scalar_decode_long(o.ctypes.data, ser[4].ctypes.data, 56); assert (o == tob([2**448 - 1])[0]).all()
    enc56 = np.zeros(56, np.uint8)
    L.goldilocks_448_scalar_encode(enc56.ctypes.data, a[9].ctypes.data); assert enc56.tobytes() == a[9].tobytes()
    assert L.goldilocks_448_scalar_eq(a[9].ctypes.data, a[9].ctypes.data) == 2**64 - 1 and L.goldilocks_448_scalar_eq(a[9].ctypes.data, a[10].ctypes.data) == 0
    L.goldilocks_448_scalar_set_unsigned(o.ctypes.data, 0xfedcba9876543210); assert o[0] == 0xfedcba9876543210 and not o[1:].any()
    L.goldilocks_448_scalar_cond_sel(o.ctypes.data, a[9].ctypes.data, a[10].ctypes.data, 0); assert (o == a[9]).all()
    L.goldilocks_448_scalar_cond_sel(o.ctypes.data, a[9].ctypes.data, a[10].ctypes.data, 5); assert (o == a[10]).all()
    L.goldilocks_448_scalar_destroy(o.ctypes.data); assert not o.any()


def _coords(points):
    """[n, 4] Python integers mod p of point_s rows (4 x 8 limbs of 56 bits, not necessarily reduced)"""
    from _libs import P
    rows = np.ascontiguousarray(points).view(np.uint64).reshape(-1, 4, 8)
    return [[sum(int(l) << (56 * i) for i, l in enumerate(c)) % P for c in r] for r in rows]


def test_debugging_torque_and_pscale_vs_oracle(ga, O):
    """goldilocks_448_point_debugging_torque / _pscale (ref: src/goldilocks.c:675-701; the reference's own tests lean on
    them, test/test_goldilocks.cxx:379-381): the coordinates the oracle gets, modulo p, lane by lane; the results are the
    same point (encodings, point_eq) in other coordinates; the drop-in names agree with the batch."""
    import torch
    from _libs import P, Point
    n = 192
    a = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-dbg"))
    fac = np.frombuffer(_gen.stream(b"t-dbg-f", 56 * n), np.uint8).reshape(n, 56).copy()
    for i, x in enumerate((0, 1, P - 1, P, P + 1, 2**448 - 1)):
        fac[i] = np.frombuffer(x.to_bytes(56, "little"), np.uint8)
    da, df = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(fac).cuda()
    out = torch.empty_like(da)
    want_t, want_s = np.empty_like(a), np.empty_like(a)
    for i in range(n):
        pin = C.cast(a[i].ctypes.data_as(C.c_void_p), C.POINTER(Point))
        O.orc_point_debugging_torque(C.cast(want_t[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)), pin)
        O.orc_point_debugging_pscale(C.cast(want_s[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)), pin, fac[i].ctypes.data_as(C.c_void_p))
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    for op, want, b in ((4, want_t, None), (5, want_s, df)):
        ga.dev("point_op", out.data_ptr(), da.data_ptr(), b.data_ptr() if b is not None else None, op, n, None)
        got = out.cpu().numpy().view(np.uint64)
        assert _coords(got) == _coords(want), op
        assert _coords(got) != _coords(a)
        assert (enc(ga, got) == enc(ga, a)).all()
        ga.dev("point_pred", st.data_ptr(), out.data_ptr(), da.data_ptr(), 0, n, None)
        assert (st.cpu().numpy() == -1).all()
    L = ga.lib()
    for i in (0, 3, 7):
        q = np.zeros(32, np.uint64)
        L.goldilocks_448_point_debugging_torque(q.ctypes.data, a[i].ctypes.data)
        assert _coords(q) == _coords(want_t[i:i + 1])
        L.goldilocks_448_point_debugging_pscale(q.ctypes.data, a[i].ctypes.data, fac[i].ctypes.data)
        assert _coords(q) == _coords(want_s[i:i + 1])


def test_precompute_matches_oracle_table(ga, O):
    from _libs import Point, Precomputed
    base = ga.point_base()
    tab = ga.precompute(base)
    assert (tab == ga.precomputed_base()).all()           # canonical limbs, bit for bit
    pt = _gen.oracle_fixed(O, _gen.random_scalars(1, b"t-pre"))[0]
    tab = ga.precompute(pt)
    want = Precomputed()
    O.orc_precompute(C.byref(want), C.cast(pt.ctypes.data_as(C.c_void_p), C.POINTER(Point)))
    assert tab.tobytes() == bytes(want)
    s = _gen.random_scalars(64, b"t-pre-s")
    got = enc(ga, ga.precomputed_scalarmul_batch(s, table=tab))
    assert (got == _gen.oracle_encode(_gen.oracle_varbase(O, np.tile(pt, (64, 1)), s))).all()


def test_verify_vs_oracle(ga, O):
    for msglen, ctx, ph in ((0, b"", False), (32, b"", False), (1, b"abc", False), (125, b"", False),
                            (126, b"", False), (200, b"x" * 255, False), (64, b"ctx", True), (300, b"", False)):
        n = 64
        sigs, pks, msgs = _gen.signatures(O, n, msglen=msglen, seed=b"t-ver%d" % msglen, nkeys=8,
                                          context=ctx, prehashed=ph)
        # corruptions: R, S, pk, message; non-canonical encodings; S >= q
        sigs[1, 3] ^= 0x10
        sigs[2, 60] ^= 0x01
        pks[3, 10] ^= 0x80
        if msglen:
            msgs[4] = bytes([msgs[4][0] ^ 1]) + msgs[4][1:]
        sigs[5, 56] |= 0x01
        pks[6, 56] |= 0x40
        sigs[7, 0:57] = 0xff
        s_val = int.from_bytes(sigs[8, 57:114].tobytes(), "little") + Q      # S + q: accepted after reduction
        sigs[8, 57:114] = np.frombuffer(s_val.to_bytes(57, "little"), np.uint8)
        got = ga.ed448_verify_batch(sigs, pks, msgs, prehashed=ph, context=ctx)
        want = _gen.oracle_verify(O, sigs, pks, msgs, context=ctx, prehashed=ph)
        assert (got == want).all(), (msglen, got, want)
        assert got[0] == -1 and got[1] == 0 and got[2] == 0 and got[8] == -1


def test_verify_ragged_batches_through_the_lane_kernel(ga, O):
    """k_ed448_verify regroups the signatures of each 256-lane block by the length of their half-size pairs
    before walking them (kernels_verify.hip): batch sizes around the wave and block boundaries, with the
    one-operation-per-wave path off so that they reach it, a third of the signatures corrupted and
    variable-length messages; every verdict must land on its own signature."""
    default = ga.get_wave_batch_max()
    ga.set_wave_batch_max(0)
    try:
        for n in (1, 2, 63, 64, 65, 255, 256, 257, 511, 1000):
            sigs, pks, msgs = _gen.signatures(O, n, msglen=20, seed=b"t-ragged-ver%d" % n, nkeys=5)
            rng = np.random.default_rng(n)
            bad = rng.random(n) < 0.33
            sigs[bad, rng.integers(0, 114, bad.sum())] ^= (1 << rng.integers(0, 8, bad.sum())).astype(np.uint8)
            got = ga.ed448_verify_batch(sigs, pks, msgs)
            want = _gen.oracle_verify(O, sigs, pks, msgs)
            assert (got == want).all(), (n, np.nonzero(got != want)[0][:8])
            assert (got[~bad] == -1).all() and (got[bad] == 0).sum() >= bad.sum() - 2
    finally:
        ga.set_wave_batch_max(default)


def test_verify_single_and_class(ga, O):
    sigs, pks, msgs = _gen.signatures(O, 2, msglen=17, seed=b"t-single")
    assert ga.ed448_verify(sigs[0].tobytes(), pks[0].tobytes(), msgs[0])
    assert not ga.ed448_verify(sigs[0].tobytes(), pks[0].tobytes(), msgs[1])
    ga.EDDSA448(pks[1].tobytes()).verify(sigs[1].tobytes(), msgs[1])
    with pytest.raises(ValueError):
        ga.EDDSA448(pks[1].tobytes()).verify(sigs[0].tobytes(), msgs[1])


def test_full_size_linearity(ga, O):
    """Size-independent property at the benchmark batch (2^20): (s1+s2)*P == s1*P + s2*P, and a sample
    of lanes against the oracle."""
    import torch
    n = 1 << 20
    rng = np.random.default_rng(7)
    k = 1 << 10
    base_k = ga.precomputed_scalarmul_batch(_gen.random_scalars(k, b"t-full-b"))
    bases = np.ascontiguousarray(base_k[rng.integers(0, k, n)])
    s1 = _gen.random_scalars(k, b"t-full-1")[rng.integers(0, k, n)]
    s2 = _gen.random_scalars(k, b"t-full-2")[rng.integers(0, k, n)]
    ssum = _gen.scalars_from_ints([int.from_bytes(a.tobytes(), "little") + int.from_bytes(b.tobytes(), "little")
                                   for a, b in zip(s1[:4096], s2[:4096])])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    db, d1, d2 = d(bases), d(s1), d(s2)
    o1, o2 = torch.empty_like(db), torch.empty_like(db)
    ga.dev("point_scalarmul", o1.data_ptr(), db.data_ptr(), d1.data_ptr(), n, None)
    ga.dev("point_scalarmul", o2.data_ptr(), db.data_ptr(), d2.data_ptr(), n, None)
    osum = torch.empty_like(db)
    ga.dev("point_op", osum.data_ptr(), o1.data_ptr(), o2.data_ptr(), 0, n, None)
    m = 4096
    dsum = d(ssum)
    o3 = torch.empty((m, 32), dtype=torch.int64, device="cuda")
    ga.dev("point_scalarmul", o3.data_ptr(), db.data_ptr(), dsum.data_ptr(), m, None)
    st = torch.empty(m, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", st.data_ptr(), o3.data_ptr(), osum.data_ptr(), 0, m, None)
    assert (st.cpu().numpy() == -1).all()
    # every output is a valid point
    stv = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", stv.data_ptr(), o1.data_ptr(), None, 1, n, None)
    assert int((stv == -1).sum()) == n
    idx = rng.integers(0, n, 256)
    got = o1.cpu().numpy().view(np.uint64)[idx]
    assert (enc(ga, got) == _gen.oracle_encode(_gen.oracle_varbase(O, bases[idx], s1[idx]))).all()


# ----------------------------------------------------------------------------- golden fixtures / KATs

import hashlib
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_f1_variable_base(ga):
    """BASELINE config 1: 1024 variable-base scalarmuls captured from the arch_ref64 build."""
    d = np.load(os.path.join(GOLD, "f1_varbase.npz"))
    bases, st = ga.point_decode_batch(d["base"], allow_identity=True)
    assert (st == -1).all()
    assert (enc(ga, ga.point_scalarmul_batch(bases, d["scalar"])) == d["out"]).all()


def test_golden_f2_fixed_base(ga):
    d = np.load(os.path.join(GOLD, "f2_fixed.npz"))
    assert (enc(ga, ga.precomputed_scalarmul_batch(d["scalar"])) == d["out"]).all()
    assert (enc(ga, ga.precomputed_scalarmul_batch(d["scalar2"], table=d["table"])) == d["out2"]).all()
    pt, st = ga.point_decode_batch(d["point"].reshape(1, 56))
    assert st[0] == -1
    assert (enc(ga, ga.precomputed_scalarmul_batch(d["scalar2"], table=ga.precompute(pt[0]))) == d["out2"]).all()


def test_golden_f3_verify(ga):
    cases = json.load(open(os.path.join(GOLD, "f3_verify.json")))["cases"]
    groups = {}
    for c in cases:
        groups.setdefault((c["ctx"], c["prehashed"]), []).append(c)
    checked = 0
    for (ctx, ph), cs in groups.items():
        sigs = np.array([np.frombuffer(bytes.fromhex(c["sig"]), np.uint8) for c in cs])
        pks = np.array([np.frombuffer(bytes.fromhex(c["pk"]), np.uint8) for c in cs])
        msgs = [bytes.fromhex(c["msg"]) for c in cs]
        got = ga.ed448_verify_batch(sigs, pks, msgs, prehashed=bool(ph), context=bytes.fromhex(ctx))
        assert list(got) == [c["verdict"] for c in cs], [c["kind"] for c in cs]
        checked += len(cs)
    assert checked == 256


def test_golden_f7_verify_torsion(ga):
    """Torsion-malleable signatures and small-order R / keys: same verdicts as the real reference."""
    cases = json.load(open(os.path.join(GOLD, "f7_verify_torsion.json")))["cases"]
    groups = {}
    for c in cases:
        groups.setdefault(c["ctx"], []).append(c)
    for ctx, cs in groups.items():
        sigs = np.array([np.frombuffer(bytes.fromhex(c["sig"]), np.uint8) for c in cs])
        pks = np.array([np.frombuffer(bytes.fromhex(c["pk"]), np.uint8) for c in cs])
        msgs = [bytes.fromhex(c["msg"]) for c in cs]
        got = ga.ed448_verify_batch(sigs, pks, msgs, context=bytes.fromhex(ctx))
        assert list(got) == [c["verdict"] for c in cs], [c["kind"] for c in cs]
    assert sum(len(v) for v in groups.values()) == 48


@pytest.mark.parametrize("keys", ["combs", "wide combs", "widest combs", "pooled tables", "every lane for itself"])
def test_golden_f3_f7_through_the_large_batch_kernels(ga, keys):
    """Large batches verify through other kernels than small ones, and how a batch's keys repeat decides which
    (kernels_verify.hip: a fixed-base comb per key without R's decoding / a pooled window table per key / every lane
    for itself).  The reference's fixtures F3 (256 cases: malformed encodings, S >= q, byte-56 rules) and F7 (48
    torsion-malleable / small-order cases, which only pass if R is compared as a point up to torsion) replicated
    into batches of 8 192 .. 16 384 signatures per (context, prehash) group, each through all three."""
    f3 = json.load(open(os.path.join(GOLD, "f3_verify.json")))["cases"]
    f7 = json.load(open(os.path.join(GOLD, "f7_verify_torsion.json")))["cases"]
    groups = {}
    for c in f3 + [dict(c, prehashed=c.get("prehashed", 0)) for c in f7]:
        groups.setdefault((c["ctx"], c["prehashed"]), []).append(c)
    try:
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT if keys != "every lane for itself" else 0, 4097)
        ga.set_verify_key_combs(ga.KEY_COMBS_DEFAULT if "combs" in keys else 0, 8)
        ga.set_verify_key_combs_wide(1 if keys == "wide combs" else 0)         # 8 teeth for every key / for none
        ga.set_verify_key_combs_xwide(1 if keys == "widest combs" else 0)      # 5 combs of 9 teeth for every key / for none
        checked = 0
        for (ctx, ph), cs in groups.items():
            reps = -(-8192 // len(cs))
            order = np.random.default_rng(len(cs)).permutation(len(cs) * reps) % len(cs)
            sigs = np.array([np.frombuffer(bytes.fromhex(c["sig"]), np.uint8) for c in cs])[order]
            pks = np.array([np.frombuffer(bytes.fromhex(c["pk"]), np.uint8) for c in cs])[order]
            msgs = [bytes.fromhex(cs[i]["msg"]) for i in order]
            got = np.asarray(ga.ed448_verify_batch(sigs, pks, msgs, prehashed=bool(ph), context=bytes.fromhex(ctx)))
            want = np.array([cs[i]["verdict"] for i in order])
            assert (got == want).all(), sorted({cs[i]["kind"] for i in order[got != want]})
            nkeys = len({c["pk"] for c in cs})                                  # ... and the batch went the way it was meant to
            assert ga.last_verify_key_counts(teeth=True) == {"combs": (nkeys, 0, nkeys, 7), "wide combs": (nkeys, 0, nkeys, 8), "widest combs": (nkeys, 0, nkeys, 9),
                                                             "pooled tables": (nkeys, nkeys, 0, 0), "every lane for itself": (0, 0, 0, 0)}[keys]
            checked += len(cs)
        assert checked == 256 + 48
    finally:
        ga.set_verify_key_pool()
        ga.set_verify_key_combs()
        ga.set_verify_key_combs_wide()
        ga.set_verify_key_combs_xwide()


@pytest.mark.parametrize("keys", ["combs", "wide combs", "widest combs", "pooled tables", "every lane for itself"])
def test_degenerate_r_and_keys_through_the_large_batch_kernels(ga, O, keys):
    """The key-comb kernel tests R without decoding it (eddsa.hpp ed448_verify_keycomb_begin): x_R = L / K with
    K = 2 Y_P (2v - u - y^2 v) v y.  The encodings where that degenerates -- y_R = 0 (K = 0: the slow path decodes R after
    all), y_R = +-1 (u = 0: the reference's isr(0) failure), y_R >= p, the sign bit flipped, byte 56 not 0 / 0x80 -- and
    the same for the keys, spliced into valid signatures of 6 keys and replicated to 12 288 signatures: every lane
    against the oracle, through all three large-batch paths."""
    P_ = 2**448 - 2**224 - 1
    enc = lambda y, s=0: np.frombuffer(int(y).to_bytes(56, "little") + bytes([0x80 * s]), np.uint8)
    sigs, pks, msgs = _gen.signatures(O, 48, msglen=20, seed=b"degenerate", nkeys=6)
    msgs = np.array([np.frombuffer(m, np.uint8) for m in msgs])
    special = [enc(0), enc(0, 1), enc(1), enc(1, 1), enc(P_ - 1), enc(P_ - 1, 1), enc(P_), enc(2**448 - 1), enc(5), enc(5, 1), enc(2), enc(P_ - 2)]
    for i, r in enumerate(special):
        sigs[2 * i, :57] = r                     # R
    sigs[25, 56] ^= 0x80                         # R's sign
    sigs[27, 56] |= 0x01                         # byte 56 of R
    sigs[29, 57:] = 0xff                         # S >= q: reduced, not rejected
    for i, r in enumerate(special[:6]):
        pks[30 + 2 * i] = r                      # the key
    pks[43, 56] ^= 0x80
    want48 = _gen.oracle_verify(O, sigs, pks, [m.tobytes() for m in msgs])
    assert (want48 == -1).sum() >= 12 and (want48 == 0).sum() >= 20
    order = np.random.default_rng(3).permutation(48 * 256) % 48
    try:
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT if keys != "every lane for itself" else 0, 4097)
        ga.set_verify_key_combs(ga.KEY_COMBS_DEFAULT if "combs" in keys else 0, 8)
        ga.set_verify_key_combs_wide(1 if keys == "wide combs" else 0)
        ga.set_verify_key_combs_xwide(1 if keys == "widest combs" else 0)
        got = np.asarray(ga.ed448_verify_batch(sigs[order], pks[order], [m.tobytes() for m in msgs[order]]))
        assert (got == want48[order]).all(), sorted(set(order[got != want48[order]]))
        distinct, pooled, combed, teeth = ga.last_verify_key_counts(teeth=True)
        assert (pooled, combed, teeth) == {"combs": (0, distinct, 7), "wide combs": (0, distinct, 8), "widest combs": (0, distinct, 9), "pooled tables": (distinct, 0, 0),
                                           "every lane for itself": (0, 0, 0)}[keys]
    finally:
        ga.set_verify_key_pool()
        ga.set_verify_key_combs()
        ga.set_verify_key_combs_wide()
        ga.set_verify_key_combs_xwide()


def test_small_batches_of_few_keys_get_combs_by_default(ga, O):
    """From the smallest batches the lane kernels see (more than 2^12 signatures) keys that repeat enough get combs
    with the library's defaults: 5 000 signatures of 16 keys, a few corrupted, every lane against the oracle."""
    n = 5_000
    sigs, pks, msgs = _gen.signatures(O, n, msglen=24, seed=b"small-combs", nkeys=16)
    sigs[::7, 60] ^= 2
    sigs[3::11, 5] ^= 1
    got = np.asarray(ga.ed448_verify_batch(sigs, pks, msgs))
    assert ga.last_verify_key_counts() == (16, 0, 16)
    want = _gen.oracle_verify(O, sigs, pks, msgs)
    assert (got == want).all() and (want == 0).sum() > n // 8 and (want == -1).sum() > n // 2


def test_rfc8032_vectors_through_the_abi(ga):
    kats = json.load(open(os.path.join(GOLD, "kats.json")))
    for c in kats["rfc8032_ed448"]:
        msg = bytes.fromhex(c["message"])
        if c["prehashed"]:
            msg = hashlib.shake_256(msg).digest(64)
        ok = ga.ed448_verify(bytes.fromhex(c["sig"]), bytes.fromhex(c["pk"]), msg,
                             context=bytes.fromhex(c["context"]), prehashed=c["prehashed"])
        assert ok
        assert not ga.ed448_verify(bytes.fromhex(c["sig"]), bytes.fromhex(c["pk"]), msg + b"x",
                                   context=bytes.fromhex(c["context"]), prehashed=c["prehashed"])
    # k*B for k < 16 (the reference's test_dalek_vectors)
    want = np.array([np.frombuffer(bytes.fromhex(h), np.uint8) for h in kats["base_multiples"]])
    ks = _gen.scalars_from_ints(list(range(16)))
    assert (enc(ga, ga.precomputed_scalarmul_batch(ks)) == want).all()
    assert (enc(ga, ga.point_scalarmul_batch(np.tile(ga.point_base(), (16, 1)), ks)) == want).all()


def test_golden_f6_full_batch_digest(ga, table_mode):
    """The whole 2^20 benchmark batch, pinned by 32 bytes computed with the real reference: under the
    digit-addressed tables bench.py's headline is quoted on, and under the library's default scan tables."""
    import torch
    dig = json.load(open(os.path.join(GOLD, "f6_bench_digest.json")))["digest_shake256_32"]
    n = 1 << 20
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    k = d(_gen.stream_scalars(n, b"bench_varbase_v1/0/base"))
    s = d(_gen.stream_scalars(n, b"bench_varbase_v1/0/scalar"))
    bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    out = torch.empty_like(bases)
    ser = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
    ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
    ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), s.data_ptr(), n, None)
    ga.dev("point_encode", ser.data_ptr(), out.data_ptr(), n, None)
    enc_all = ser.cpu().numpy()
    for lg in (10, 16, 20):
        assert hashlib.shake_256(enc_all[:1 << lg].tobytes()).hexdigest(32) == dig[str(lg)], lg


# ----------------------------------------------------------------------------- "next" rows


def test_derive_public_key_and_sign_vs_oracle_and_rfc8032(ga, O):
    kats = json.load(open(os.path.join(GOLD, "kats.json")))["rfc8032_ed448"]
    for c in kats:
        sk, pk, ctx = bytes.fromhex(c["sk"]), bytes.fromhex(c["pk"]), bytes.fromhex(c["context"])
        msg = bytes.fromhex(c["message"])
        if c["prehashed"]:
            msg = hashlib.shake_256(msg).digest(64)
        sk_a, pk_a = np.frombuffer(sk, np.uint8).reshape(1, 57), np.frombuffer(pk, np.uint8).reshape(1, 57)
        assert ga.ed448_derive_public_key_batch(sk_a)[0].tobytes() == pk
        sig = ga.ed448_sign_batch(sk_a, pk_a, [msg], prehashed=c["prehashed"], context=ctx)
        assert sig[0].tobytes().hex() == c["sig"]
    # seeded batch vs the oracle's signer, ragged message lengths, then verify on the GPU
    n = 200
    sk = np.frombuffer(_gen.stream(b"t-sign-sk", 57 * n), np.uint8).reshape(n, 57).copy()
    want_pk = np.empty((n, 57), np.uint8)
    O.orc_ed448_derive_public_key_batch(want_pk.ctypes.data, sk.ctypes.data, n, 8)
    pk = ga.ed448_derive_public_key_batch(sk)
    assert (pk == want_pk).all()
    msgs = [_gen.stream(b"t-sign-m%d" % i, 400)[:(i * 7) % 311] for i in range(n)]
    for ctx, ph in ((b"", False), (b"ctx!", False), (b"", True)):
        sig = ga.ed448_sign_batch(sk, pk, msgs, prehashed=ph, context=ctx)
        for i in range(0, n, 3):
            w = (C.c_uint8 * 114)()
            m = (C.c_uint8 * max(1, len(msgs[i]))).from_buffer_copy(msgs[i] or b"\0")
            cb = (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0")
            O.orc_ed448_sign(w, sk[i].ctypes.data, pk[i].ctypes.data, m, len(msgs[i]), 1 if ph else 0, cb, len(ctx))
            assert bytes(w) == sig[i].tobytes(), (i, ctx, ph)
        assert (ga.ed448_verify_batch(sig, pk, msgs, prehashed=ph, context=ctx) == -1).all()


def test_direct_scalarmul_wire_format(ga, O):
    n = 300
    s = _gen.random_scalars(n, b"t-direct-s")
    pts = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-direct-b"))
    base = _gen.oracle_encode(pts)
    base[5] = 0                       # identity encoding
    base[6] = 0xff                    # not a field element
    base[7, 0] |= 1                   # negative s
    for allow_id in (False, True):
        for short in (False, True):
            got, st = ga.direct_scalarmul_batch(base, s, allow_identity=allow_id, short_circuit=short)
            for i in list(range(12)) + list(range(12, n, 17)):
                out = (C.c_uint8 * 56)()
                from _libs import Scalar
                r = O.orc_direct_scalarmul(out, base[i].ctypes.data, C.cast(s[i].ctypes.data, C.POINTER(Scalar)),
                                           1 if allow_id else 0, 1 if short else 0)
                assert r == st[i], (i, allow_id, short)
                if r == -1 or not short:
                    assert bytes(out) == got[i].tobytes(), (i, allow_id, short)
                else:
                    assert not got[i].any()          # untouched (the binding passes zeros in)
    one = np.frombuffer(bytes(base[0]), np.uint8)
    out = (C.c_uint8 * 56)()
    r = ga.lib().goldilocks_448_direct_scalarmul(out, one.ctypes.data, s[0].ctypes.data, 0, 0)
    assert r == -1 and bytes(out) == ga.direct_scalarmul_batch(base[:1], s[:1])[0][0].tobytes()


def test_x448_vs_oracle_and_rfc7748(ga, O):
    kats = json.load(open(os.path.join(GOLD, "kats.json")))["rfc7748_x448_iterated"]
    u = k = np.frombuffer(bytes([5] + [0] * 55), np.uint8).reshape(1, 56)
    for i in range(1000):                       # the reference's iterated test (test_goldilocks.cxx:545-552)
        out, st = ga.x448_batch(k, u)
        assert st[0] == -1
        u, k = k, out
        if i == 0:
            assert k[0].tobytes().hex() == kats["1"]
    assert k[0].tobytes().hex() == kats["1000"]
    n = 512
    sc = np.frombuffer(_gen.stream(b"t-x448-s", 56 * n), np.uint8).reshape(n, 56).copy()
    bs = np.frombuffer(_gen.stream(b"t-x448-b", 56 * n), np.uint8).reshape(n, 56).copy()
    bs[0] = 0                     # low-order input: result must be zero -> FAILURE
    bs[1] = 0xff
    bs[2] = 0; bs[2, 0] = 1
    for row, val in zip(range(4, 9), (P - 1, P, P + 1, 2, 5)):        # more low-order / non-canonical u
        bs[row] = np.frombuffer(val.to_bytes(56, "little"), np.uint8)
    got, st = ga.x448_batch(sc, bs)
    pub, _ = ga.x448_batch(sc)
    for i in range(n):
        w = (C.c_uint8 * 56)()
        assert O.orc_x448(w, bs[i].ctypes.data, sc[i].ctypes.data) == st[i], i
        assert bytes(w) == got[i].tobytes(), i
        O.orc_x448_derive_public_key(w, sc[i].ctypes.data)
        assert bytes(w) == pub[i].tobytes(), i
    assert st[0] == 0 and st[3] == -1
    # Diffie-Hellman consistency through the drop-in single-op names
    L = ga.lib()
    a, b = sc[10].copy(), sc[11].copy()
    pa, pb, s1, s2 = (np.empty(56, np.uint8) for _ in range(4))
    L.goldilocks_x448_derive_public_key(pa.ctypes.data, a.ctypes.data)
    L.goldilocks_x448_derive_public_key(pb.ctypes.data, b.ctypes.data)
    assert L.goldilocks_x448(s1.ctypes.data, pb.ctypes.data, a.ctypes.data) == -1
    assert L.goldilocks_x448(s2.ctypes.data, pa.ctypes.data, b.ctypes.data) == -1
    assert (s1 == s2).all()


def test_x448_conversions_vs_oracle(ga, O):
    """The rest of the X448 surface (src/goldilocks.c:1079-1115, src/eddsa.c:83-128): Ed448 public / private keys to X448,
    the secret scalar, points encoded like X448 -- against the oracle (pinned to the reference by
    test_oracle_vs_ref.py::test_x448_conversions_differential), over a ragged batch and through the single-call names;
    and the reference's own check (test_goldilocks.cxx:625-655): converting a public key == deriving from the converted
    private key."""
    from _libs import Point, Scalar
    n = 700                                    # ragged: 10 full waves + 60 lanes
    sk = np.frombuffer(_gen.stream(b"t-conv-sk", 57 * n), np.uint8).reshape(n, 57).copy()
    pk = ga.ed448_derive_public_key_batch(sk)
    ed = pk.copy()
    for row, val in zip(range(6), (0, 1, P - 1, P, P + 1, 2**448 - 1)):     # 1/(1 - y^2) = 1/0; y >= p taken as it stands
        ed[row, :56] = np.frombuffer(val.to_bytes(56, "little"), np.uint8)
    xpub, xpriv, secret = (ga.x448_from_edwards_batch(k, a) for k, a in (("public", ed), ("private", sk), ("scalar", sk)))
    pts = ga.point_from_hash_batch(np.frombuffer(_gen.stream(b"t-conv-pt", 112 * n), np.uint8).reshape(n, 112), uniform=True).view(np.uint8).reshape(n, 256)
    pts[0] = np.frombuffer(bytes(Point.in_dll(ga.lib(), "goldilocks_448_point_identity")), np.uint8)   # x = 0
    like = ga.x448_from_edwards_batch("point", pts)
    w, s = (C.c_uint8 * 56)(), Scalar()
    for i in range(n):
        O.orc_ed448_convert_public_key_to_x448(w, ed[i].ctypes.data)
        assert bytes(w) == xpub[i].tobytes(), i
        O.orc_ed448_convert_private_key_to_x448(w, sk[i].ctypes.data)
        assert bytes(w) == xpriv[i].tobytes(), i
        O.orc_ed448_derive_secret_scalar(C.byref(s), sk[i].ctypes.data)
        assert bytes(s) == secret[i].tobytes(), i
        O.orc_point_encode_like_x448(w, pts[i].ctypes.data)
        assert bytes(w) == like[i].tobytes(), i
    derived, _ = ga.x448_batch(xpriv[6:])
    assert (derived == xpub[6:]).all()
    L = ga.lib()
    one = np.empty(56, np.uint8)
    for name, src, want in (("goldilocks_ed448_convert_public_key_to_x448", ed[9], xpub[9]),
                            ("goldilocks_ed448_convert_private_key_to_x448", sk[9], xpriv[9]),
                            ("goldilocks_ed448_derive_secret_scalar", sk[9], secret[9]),
                            ("goldilocks_448_point_mul_by_ratio_and_encode_like_x448", pts[9], like[9])):
        getattr(L, name)(one.ctypes.data, src.ctypes.data)
        assert (one == want).all(), name


def test_elligator_and_dual_scalarmul(ga, O):
    from _libs import Point
    kats = json.load(open(os.path.join(GOLD, "kats.json")))["elligator_nonuniform"]
    h = np.array([np.frombuffer(bytes.fromhex(c["hash"]), np.uint8) for c in kats])
    want = np.array([np.frombuffer(bytes.fromhex(c["point"]), np.uint8) for c in kats])
    assert (enc(ga, ga.point_from_hash_batch(h)) == want).all()          # the reference's elligator_examples
    n = 300
    raw = np.frombuffer(_gen.stream(b"t-elligator", 112 * n), np.uint8).reshape(n, 112).copy()
    raw[0] = 0
    raw[1] = 0xff
    for uniform in (False, True):
        got = ga.point_from_hash_batch(raw if uniform else raw[:, :56].copy(), uniform=uniform)
        w = np.empty((n, 32), np.uint64)
        for i in range(n):
            f = O.orc_point_from_hash_uniform if uniform else O.orc_point_from_hash_nonuniform
            f(C.cast(w[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)), raw[i].ctypes.data)
        assert (enc(ga, got) == _gen.oracle_encode(w)).all(), uniform
    pts = ga.point_from_hash_batch(raw, uniform=True)
    s1, s2 = _gen.random_scalars(n, b"t-dual-1"), _gen.random_scalars(n, b"t-dual-2")
    o1, o2 = ga.point_dual_scalarmul_batch(pts, s1, s2)
    assert (enc(ga, o1) == _gen.oracle_encode(_gen.oracle_varbase(O, pts, s1))).all()
    assert (enc(ga, o2) == _gen.oracle_encode(_gen.oracle_varbase(O, pts, s2))).all()


def test_long_messages_and_contexts(ga, O):
    """Multi-block SHAKE absorption: messages of 0 .. 100 000 bytes (lane-divergent lengths in one
    batch) with a 255-byte context, signed and verified on the GPU, byte-identical to the oracle."""
    lens = [0, 1, 20, 21, 22, 135, 136, 137, 271, 272, 273, 1000, 4095, 4096, 65537, 100000]
    n = len(lens)
    sk = np.frombuffer(_gen.stream(b"long/sk", 57 * n), np.uint8).reshape(n, 57).copy()
    msgs = [_gen.stream(b"long/msg/%d" % i, l) if l else b"" for i, l in enumerate(lens)]
    ctx = bytes(range(255))
    pk = ga.ed448_derive_public_key_batch(sk)
    sig = ga.ed448_sign_batch(sk, pk, msgs, context=ctx)
    assert (ga.ed448_verify_batch(sig, pk, msgs, context=ctx) == -1).all()
    cctx = (C.c_uint8 * 255).from_buffer_copy(ctx)
    for i in range(n):
        w = (C.c_uint8 * 114)()
        mb = (C.c_uint8 * max(1, lens[i])).from_buffer_copy(msgs[i] or b"\0")
        O.orc_ed448_sign(w, sk[i].ctypes.data, pk[i].ctypes.data, mb, lens[i], 0, cctx, 255)
        assert bytes(w) == sig[i].tobytes(), lens[i]
    bad = [m[:-1] + bytes([m[-1] ^ 1]) if m else b"x" for m in msgs]
    assert (ga.ed448_verify_batch(sig, pk, bad, context=ctx) == 0).all()


@pytest.mark.parametrize("wave_path", [True, False])
def test_device_offsets_with_an_oversize_message(ga, O, wave_path):
    """Messages behind device-resident offsets (msg_offsets != NULL): msg_len is ignored there, so a garbage
    msg_len must not fail the call; and a lane whose offsets claim GOLDILOCKS_AMD_MAX_MESSAGE_BYTES or more cannot
    be hashed by the 32-bit byte counters -- signing writes an all-zero signature for it and verification rejects
    it (include/goldilocks_amd.h), the other lanes are untouched.  The oversize message itself is never read."""
    import torch
    n = 4
    saved = ga.get_wave_batch_max()
    ga.set_wave_batch_max(saved if wave_path else 0)
    try:
        sk = np.frombuffer(_gen.stream(b"oversize/sk", 57 * n), np.uint8).reshape(n, 57).copy()
        pk = ga.ed448_derive_public_key_batch(sk)
        msgs = [b"abc", b"", b"0123456789" * 3, b"x"]
        order = [0, 2, 3, 1]                                   # the oversize lane last: the others keep real bytes
        blob = np.frombuffer(b"".join(msgs[i] for i in order[:3]), np.uint8).copy()
        off = np.zeros(n + 1, np.uint64)
        off[1:4] = np.cumsum([len(msgs[i]) for i in order[:3]])
        off[4] = off[3] + np.uint64(0x80000000)              # the last lane: 2^31 "bytes"
        sk_o, pk_o = sk[order], pk[order]
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d_sk, d_pk, d_blob = d(sk_o), d(pk_o), d(blob)
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        sig = torch.full((n, 114), 0x55, dtype=torch.uint8, device="cuda")
        garbage_len = 2**64 - 1                                # ignored when offsets are given
        ga.dev("ed448_sign", sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_blob.data_ptr(), d_off.data_ptr(),
               garbage_len, 0, None, 0, n, None)
        st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), d_pk.data_ptr(), d_blob.data_ptr(), d_off.data_ptr(),
               garbage_len, 0, None, 0, n, None)
        torch.cuda.synchronize()
        sig_h, st_h = sig.cpu().numpy(), st.cpu().numpy()
        assert (sig_h[3] == 0).all() and list(st_h) == [-1, -1, -1, 0]
        want = ga.ed448_sign_batch(sk_o[:3], pk_o[:3], [msgs[i] for i in order[:3]])
        assert (sig_h[:3] == want).all()
        assert (_gen.oracle_verify(O, sig_h[:3], pk_o[:3], [msgs[i] for i in order[:3]]) == -1).all()
    finally:
        ga.set_wave_batch_max(saved)
