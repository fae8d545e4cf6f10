"""GPU parity of the one-operation-per-WAVE path (libgoldilocks_amd/csrc/wave_coop.hpp): batches of up to
goldilocks_amd_get_wave_batch_max() variable-base multiplications -- the single-operation drop-in name
included -- run with the 64 lanes of a wavefront sharing one operation (0.35-0.6 ms instead of the
2.1-2.5 ms of a lane's ladder).  Same reference lines as the lane kernels (src/goldilocks.c:405-465),
same bar: bit-exact encodings, against the reference's golden vectors (F1) and the oracle."""
import os

import numpy as np
import pytest

import _gen
from _libs import Q

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def paths(ga):
    """Run a body through the wave path (threshold raised) and the lane path (threshold 0)."""
    default = ga.get_wave_batch_max()

    def both(fn):
        out = {}
        try:
            for name, mx in (("wave", 1 << 20), ("lane", 0)):
                ga.set_wave_batch_max(mx)
                out[name] = fn()
        finally:
            ga.set_wave_batch_max(default)
        return out
    return both


def test_default_threshold_is_the_documented_one(ga):
    assert ga.get_wave_batch_max() == 8192


def test_golden_f1_through_the_wave_path(ga, paths):
    d = np.load(os.path.join(G, "f1_varbase.npz"))
    bases, st = ga.point_decode_batch(d["base"], allow_identity=True)
    assert (st == -1).all()
    r = paths(lambda: ga.point_encode_batch(ga.point_scalarmul_batch(bases, d["scalar"])))
    assert (r["wave"] == d["out"]).all()
    assert (r["lane"] == d["out"]).all()


def test_ragged_batches_edge_scalars_and_special_points(ga, O, paths):
    edge = [0, 1, 2, Q - 1, Q - 2, 2**445, 2**445 - 1, (Q + 1) // 2, 15, 16, 17, 31, 32, 33, 2**224, 2**440 + 12345]
    for n in (1, 2, 3, 5, 63, 64, 65, 257, 1000):
        k = _gen.stream_scalars(n, b"wave/base/%d" % n)
        bases = _gen.oracle_fixed(O, k)
        s = _gen.stream_scalars(n, b"wave/scalar/%d" % n)
        m = min(n, len(edge))
        s[:m] = _gen.scalars_from_ints(edge[:m])
        if n >= 5:
            bases[3] = ga.point_identity()             # the identity as base
            bases[4] = ga.point_base()
        want = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))
        r = paths(lambda: ga.point_encode_batch(ga.point_scalarmul_batch(bases, s)))
        assert (r["wave"] == want).all(), n
        assert (r["lane"] == want).all(), n


def test_unreduced_and_rescaled_representatives(ga, O, paths):
    """Weakly reduced limbs (above 2^56) and projectively rescaled inputs give the same group element."""
    from _libs import P
    n = 64
    bases = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"wave/rep/base"))
    s = _gen.stream_scalars(n, b"wave/rep/scalar")
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))
    rng = np.random.default_rng(3)
    alt = bases.copy()
    pl = [(1 << 56) - 1] * 8
    pl[4] -= 1
    for i in range(n):
        c = [sum(int(x) << (56 * j) for j, x in enumerate(bases[i, 8 * k:8 * k + 8])) % P for k in range(4)]
        f = int.from_bytes(rng.bytes(56), "little") % P or 1
        c = [x * f % P for x in c]
        limbs = []
        for x in c:
            l = [(x >> (56 * j)) & ((1 << 56) - 1) for j in range(8)]
            if i % 2:
                l = [a + b for a, b in zip(l, pl)]       # + p limb-wise: limbs above 2^56
            limbs += l
        alt[i] = np.array(limbs, dtype=np.uint64)
    r = paths(lambda: ga.point_encode_batch(ga.point_scalarmul_batch(alt, s)))
    assert (r["wave"] == want).all() and (r["lane"] == want).all()


def test_single_operation_drop_in_and_in_place_output(ga, O):
    """goldilocks_448_point_scalarmul (one operation) takes the wave path by default; out may alias base."""
    import torch
    k = _gen.stream_scalars(4, b"wave/single/base")
    bases = _gen.oracle_fixed(O, k)
    s = _gen.stream_scalars(4, b"wave/single/scalar")
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))
    for i in range(4):
        got = ga.point_scalarmul(bases[i], s[i])
        assert (ga.point_encode_batch(got.reshape(1, 32)) == want[i]).all()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    io, ds = d(bases), d(s)
    ga.dev("point_scalarmul", io.data_ptr(), io.data_ptr(), ds.data_ptr(), 4, None)      # in place
    torch.cuda.synchronize()
    assert (ga.point_encode_batch(io.cpu().numpy().view(np.uint64)) == want).all()


def test_both_sides_of_the_threshold_agree(ga, O):
    default = ga.get_wave_batch_max()
    try:
        ga.set_wave_batch_max(100)
        k = _gen.stream_scalars(101, b"wave/thr/base")
        bases = _gen.oracle_fixed(O, k)
        s = _gen.stream_scalars(101, b"wave/thr/scalar")
        a = ga.point_encode_batch(ga.point_scalarmul_batch(bases[:100], s[:100]))      # wave
        b = ga.point_encode_batch(ga.point_scalarmul_batch(bases, s))                  # lane (101 > 100)
        assert (a == b[:100]).all()
        assert (b == _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))).all()
    finally:
        ga.set_wave_batch_max(default)


def test_verification_through_both_paths(ga, O, paths):
    """goldilocks_ed448_verify, one verification per wave against one per lane: valid, corrupted (R, S, key,
    message), contexts, ragged message lengths, RFC 8032's vectors and the torsion fixture F7."""
    import json
    sigs, pks, msgs = _gen.signatures(O, 300, msglen=40, seed=b"wave/verify", nkeys=17, context=b"ctx")
    rng = np.random.default_rng(9)
    for i in range(0, 300, 3):
        which = i % 4
        if which == 0: sigs[i, rng.integers(0, 57)] ^= 1 << rng.integers(0, 8)
        elif which == 1: sigs[i, 57 + rng.integers(0, 56)] ^= 1 << rng.integers(0, 8)
        elif which == 2: pks[i, rng.integers(0, 56)] ^= 1 << rng.integers(0, 8)
        else: msgs[i] = msgs[i][:-1] + bytes([msgs[i][-1] ^ 1])
    msgs = [m[:len(m) - (i % 7)] if i % 5 == 0 else m for i, m in enumerate(msgs)]      # ragged lengths (and thus invalid)
    want = _gen.oracle_verify(O, sigs, pks, msgs, context=b"ctx")
    r = paths(lambda: ga.ed448_verify_batch(sigs, pks, msgs, context=b"ctx"))
    assert (r["wave"] == want).all() and (r["lane"] == want).all()
    assert 0 < (want == -1).sum() < 300
    groups = {}
    for c in json.load(open(os.path.join(G, "f7_verify_torsion.json")))["cases"]:
        groups.setdefault(c["ctx"], []).append(c)
    for ctx, cs in groups.items():
        f = lambda k: np.array([np.frombuffer(bytes.fromhex(c[k]), np.uint8) for c in cs])
        fm = [bytes.fromhex(c["msg"]) for c in cs]
        r = paths(lambda: ga.ed448_verify_batch(f("sig"), f("pk"), fm, context=bytes.fromhex(ctx)))
        verdicts = np.array([c["verdict"] for c in cs])
        assert (r["wave"] == verdicts).all() and (r["lane"] == verdicts).all()


def test_wave_field_arithmetic_against_exact_integers(ga):
    """The row arithmetic itself (goldilocks_amd_wave_field_op_dev): mul, strong_reduce, isr, eq, lobit,
    deserialize, on special values, unreduced limbs and random elements, every element checked."""
    import torch
    from _libs import P
    M = (1 << 56) - 1
    val = lambda l: sum(int(x) << (56 * i) for i, x in enumerate(l)) % P
    raw = lambda l: sum(int(x) << (56 * i) for i, x in enumerate(l))

    def run(op, a, b=None):
        n = len(a)
        da = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()
        db = da if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint64).view(np.int64)).cuda()
        out = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
        st = torch.zeros(n, dtype=torch.int32, device="cuda")
        ga.dev("wave_field_op", out.data_ptr(), st.data_ptr(), da.data_ptr(), db.data_ptr(), op, n, None)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint64), st.cpu().numpy()
    rng = np.random.default_rng(1)
    n = 403                                                          # not a multiple of 4: a partial last wave
    a = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    b = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    a[0] = 0; a[1] = M; a[2] = [M] * 4 + [M - 1] + [M] * 3; a[3] = [M - 1] + [M] * 3 + [M - 1] + [M] * 3
    a[4] = [1, 0, 0, 0, 0, 0, 0, 0]; a[5] = [2**56 + 255] * 8; b[1] = M; b[5] = [2**56 + 255] * 8
    o, _ = run(0, a, b)
    assert all(val(o[i]) == val(a[i]) * val(b[i]) % P for i in range(n))
    o, _ = run(1, a)
    assert all(raw(o[i]) == val(a[i]) for i in range(n))
    o, s = run(2, a[:64])
    for i in range(64):
        x = val(a[i]); r = pow(x, (P - 3) // 4, P)
        assert val(o[i]) == r and (s[i] != 0) == (r * r * x % P == 1)
    b2 = a.copy(); b2[::3, 0] ^= np.uint64(1); b2[1::5] += np.array([M] * 4 + [M - 1] + [M] * 3, dtype=np.uint64)   # + p
    _, s = run(3, a, b2)
    assert all((s[i] != 0) == (val(a[i]) == val(b2[i])) for i in range(n))
    _, s = run(4, a)
    assert all((s[i] != 0) == bool(val(a[i]) & 1) for i in range(n))
    vals = [0, 1, P - 1, P, P + 1, 2**448 - 1, 2**447] + [int.from_bytes(rng.bytes(56), "little") for _ in range(90)]
    rb = np.zeros((len(vals), 8), dtype=np.uint64)
    for i, v in enumerate(vals):
        rb[i, :7] = np.frombuffer(v.to_bytes(56, "little"), dtype=np.uint64)
    o, s = run(5, rb)
    assert all(raw(o[i]) == vals[i] and (s[i] != 0) == (vals[i] < P) for i in range(len(vals)))


def test_double_base_multiplications_through_both_paths(ga, O, paths):
    """s1*P1 + s2*P2 and s1*B + s2*P2 (what the reference's eddsa.c verification calls), ragged sizes."""
    import ctypes as C
    from _libs import Point, Scalar
    pt = lambda a: a.ctypes.data_as(C.POINTER(Point))
    sc = lambda a: a.ctypes.data_as(C.POINTER(Scalar))
    for n in (1, 5, 130):
        b1 = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"wave/dbl/b1/%d" % n))
        b2 = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"wave/dbl/b2/%d" % n))
        s1 = _gen.stream_scalars(n, b"wave/dbl/s1/%d" % n)
        s2 = _gen.stream_scalars(n, b"wave/dbl/s2/%d" % n)
        s1[0] = 0
        s2[n - 1] = _gen.scalars_from_ints([Q - 1])[0]
        w1, w2 = np.empty((n, 32), np.uint64), np.empty((n, 32), np.uint64)
        for i in range(n):
            O.orc_point_double_scalarmul(pt(w1[i]), pt(b1[i]), sc(s1[i]), pt(b2[i]), sc(s2[i]))
            O.orc_base_double_scalarmul_non_secret(pt(w2[i]), sc(s1[i]), pt(b2[i]), sc(s2[i]))
        e1, e2 = _gen.oracle_encode(w1), _gen.oracle_encode(w2)
        r = paths(lambda: (ga.point_encode_batch(ga.point_double_scalarmul_batch(b1, s1, b2, s2)),
                           ga.point_encode_batch(ga.point_double_scalarmul_batch(None, s1, b2, s2))))
        for name in ("wave", "lane"):
            assert (r[name][0] == e1).all() and (r[name][1] == e2).all(), (name, n)


def test_x448_through_both_paths(ga, O, paths):
    """goldilocks_x448 with a peer's point: one Montgomery ladder per wave against one per lane, random
    inputs, low-order and non-canonical u-coordinates (results 0 -> FAILURE), ragged sizes."""
    import ctypes as C
    from _libs import P
    special = [0, 1, P - 1, P, P + 1, 2**448 - 1, 5]
    for n in (1, 7, 200):
        sc_ = np.frombuffer(_gen.stream(b"wave/x448/s/%d" % n, 56 * n), np.uint8).reshape(n, 56).copy()
        u = np.frombuffer(_gen.stream(b"wave/x448/u/%d" % n, 56 * n), np.uint8).reshape(n, 56).copy()
        for i, v in enumerate(special[:n]):
            u[i] = np.frombuffer(v.to_bytes(56, "little"), np.uint8)
        want, want_st = np.empty((n, 56), np.uint8), np.empty(n, np.int32)
        for i in range(n):
            w = (C.c_uint8 * 56)()
            want_st[i] = O.orc_x448(w, u[i].ctypes.data_as(C.c_void_p), sc_[i].ctypes.data_as(C.c_void_p))
            want[i] = np.frombuffer(bytes(w), np.uint8)
        r = paths(lambda: ga.x448_batch(sc_, u))
        for name in ("wave", "lane"):
            got, st = r[name]
            assert (st == want_st).all(), (name, n)
            assert (got == want).all(), (name, n)


def test_fixed_base_keygen_and_signing_through_both_paths(ga, O, paths):
    """precomputed_scalarmul (built-in and caller table), ed448 derive_public_key / sign, x448 key generation:
    one operation per wave against the lane kernels and the oracle, RFC 8032's vectors included."""
    import ctypes as C
    import hashlib
    import json
    _p = lambda a: a.ctypes.data_as(C.c_void_p)
    for n in (1, 6, 150):
        k = _gen.stream_scalars(n, b"wave/fixed/k/%d" % n)
        k[0] = 0
        want = _gen.oracle_encode(_gen.oracle_fixed(O, k))
        other = ga.precompute(_gen.oracle_fixed(O, _gen.stream_scalars(1, b"wave/fixed/pt"))[0])
        want2 = _gen.oracle_encode(_gen.oracle_fixed(O, k, table=other))
        sk = np.frombuffer(_gen.stream(b"wave/fixed/sk/%d" % n, 57 * n), np.uint8).reshape(n, 57).copy()
        xs = np.frombuffer(_gen.stream(b"wave/fixed/xs/%d" % n, 56 * n), np.uint8).reshape(n, 56).copy()
        msgs = [_gen.stream(b"wave/fixed/msg/%d/%d" % (n, i), (i * 37) % 300) for i in range(n)]      # ragged, up to 3 blocks
        want_pk = np.empty((n, 57), np.uint8)
        O.orc_ed448_derive_public_key_batch(_p(want_pk), _p(sk), n, _gen.NTHREADS)
        want_sig, want_x = np.empty((n, 114), np.uint8), np.empty((n, 56), np.uint8)
        ctx = (C.c_uint8 * 3).from_buffer_copy(b"abc")
        for i in range(n):
            m = (C.c_uint8 * max(1, len(msgs[i]))).from_buffer_copy(msgs[i] or b"\0")
            O.orc_ed448_sign(_p(want_sig[i]), _p(sk[i]), _p(want_pk[i]), m, len(msgs[i]), 0, ctx, 3)
            O.orc_x448_derive_public_key(_p(want_x[i]), _p(xs[i]))

        def body():
            pk = ga.ed448_derive_public_key_batch(sk)
            return dict(fixed=ga.point_encode_batch(ga.precomputed_scalarmul_batch(k)),
                        fixed2=ga.point_encode_batch(ga.precomputed_scalarmul_batch(k, table=other)),
                        pk=pk, sig=ga.ed448_sign_batch(sk, pk, msgs, context=b"abc"), x=ga.x448_batch(xs)[0])
        for name, r in paths(body).items():
            assert (r["fixed"] == want).all() and (r["fixed2"] == want2).all(), (name, n)
            assert (r["pk"] == want_pk).all(), (name, n)
            assert (r["sig"] == want_sig).all(), (name, n)
            assert (r["x"] == want_x).all(), (name, n)
    for c in json.load(open(os.path.join(G, "kats.json")))["rfc8032_ed448"]:
        msg = bytes.fromhex(c["message"])
        if c["prehashed"]:
            msg = hashlib.shake_256(msg).digest(64)
        sk1 = np.frombuffer(bytes.fromhex(c["sk"]), np.uint8).reshape(1, 57)
        r = paths(lambda: (ga.ed448_derive_public_key_batch(sk1),
                           ga.ed448_sign_batch(sk1, np.frombuffer(bytes.fromhex(c["pk"]), np.uint8).reshape(1, 57), [msg],
                                               prehashed=bool(c["prehashed"]), context=bytes.fromhex(c["context"]))))
        for name in ("wave", "lane"):
            assert r[name][0].tobytes().hex() == c["pk"] and r[name][1].tobytes().hex() == c["sig"], name


def test_precompute_by_one_wave_per_table(ga, O):
    """goldilocks_448_precompute for a handful of points through both paths (one table per wave; one table per
    lane with the wave path off): 15 360 bytes each, identical to the oracle's table -- which is the reference's."""
    import ctypes as C
    import torch
    from _libs import Point, Precomputed
    n = 9
    pts = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"wave/precompute"))
    pts[0] = ga.point_base()
    want = []
    for i in range(n):
        w = Precomputed()
        O.orc_precompute(C.byref(w), C.cast(pts[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)))
        want.append(bytes(w))
    dp = torch.from_numpy(pts.view(np.int64)).cuda()
    default = ga.get_wave_batch_max()
    try:
        for mx in (default, 0):
            ga.set_wave_batch_max(mx)
            tabs = torch.zeros((n, 1920), dtype=torch.int64, device="cuda")
            ga.dev("precompute", tabs.data_ptr(), dp.data_ptr(), n, None)
            got = tabs.cpu().numpy()
            for i in range(n):
                assert got[i].tobytes() == want[i], (mx, i)
    finally:
        ga.set_wave_batch_max(default)
    assert want[0] == ga.precomputed_base().tobytes()
