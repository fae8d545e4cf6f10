"""Larger randomized differential runs of every kernel against the oracle (rare-carry hunting):
32768 lanes per kernel, bit-exact on encodings / bytes / status.  GOLDILOCKS_SOAK_SEED=<text> draws a
different input stream (used for extended soaks on the GPU box; the default stream is fixed)."""
import ctypes as C

import numpy as np
import pytest

import _gen
from _libs import Q

pytestmark = pytest.mark.gpu
N = 1 << 15
import os
SEED = os.environ.get("GOLDILOCKS_SOAK_SEED", "").encode()


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


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def enc(ga, pts):
    """the GPU's raw points encoded by the device AND by the oracle's orc_point_encode: equal, or the device's encoder
    is what differs"""
    got = ga.point_encode_batch(pts)
    assert (got == _gen.oracle_encode(pts)).all()
    return got


def reference_check(bases, scalars, got_enc, k=64):
    """a sample of lanes against the REAL reference (oracle/_ref/libgoldilocks_ref64.so, arch_ref64 compiled from
    /root/reference by oracle/Makefile; it travels to the GPU box with the snapshot): goldilocks_448_point_scalarmul
    (src/goldilocks.c:405-465) + goldilocks_448_point_encode on the same inputs"""
    from _libs import Point, Scalar, have_ref, ref
    if not have_ref():
        return 0
    R = ref()
    pick = np.random.default_rng(7).choice(len(bases), k, replace=False)
    for i in pick:
        b, s, out = Point(), Scalar(), Point()
        C.memmove(C.byref(b), bases[i].ctypes.data, 256)
        C.memmove(C.byref(s), scalars[i].ctypes.data, 56)
        R.goldilocks_448_point_scalarmul(C.byref(out), C.byref(b), C.byref(s))
        e = (C.c_uint8 * 56)()
        R.goldilocks_448_point_encode(e, C.byref(out))
        assert bytes(e) == got_enc[i].tobytes(), i
    return k


def test_soak_scalarmuls(ga, O):
    k = _gen.stream_scalars(N, SEED + b"soak/base")
    s = _gen.stream_scalars(N, SEED + b"soak/scalar")
    bases = ga.precomputed_scalarmul_batch(k)                                     # 8-bit window table
    assert (enc(ga, bases) == _gen.oracle_encode(_gen.oracle_fixed(O, k))).all()
    comb = ga.precomputed_scalarmul_batch(k, table=ga.precomputed_base())         # LDS comb
    assert (ga.point_encode_batch(comb) == ga.point_encode_batch(bases)).all()
    got = ga.point_scalarmul_batch(bases, s)
    got_enc = enc(ga, got)
    assert (got_enc == _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))).all()
    reference_check(bases, s, got_enc)
    e57 = ga.point_encode_like_eddsa_batch(got)
    dec, st = ga.point_decode_like_eddsa_batch(e57)
    four = ga.point_scalarmul_batch(got, _gen.scalars_from_ints([4] * N))
    ok = st == -1        # the encoding of the identity (4P = 0) does not decode, as in the reference
    assert ok.sum() >= N - 4 and (ga.point_encode_batch(dec)[ok] == ga.point_encode_batch(four)[ok]).all()


def test_soak_sign_verify(ga, O):
    sk = np.frombuffer(_gen.stream(SEED + b"soak/sk", 57 * N), np.uint8).reshape(N, 57).copy()
    pk = ga.ed448_derive_public_key_batch(sk)
    want_pk = np.empty_like(pk)
    O.orc_ed448_derive_public_key_batch(_p(want_pk), _p(sk), N, _gen.NTHREADS)
    assert (pk == want_pk).all()
    msg = np.frombuffer(_gen.stream(SEED + b"soak/msg", 48 * N), np.uint8).reshape(N, 48).copy()
    msgs = [m.tobytes() for m in msg]
    sig = ga.ed448_sign_batch(sk, pk, msgs, context=b"soak")
    want = np.empty_like(sig)
    ctx = (C.c_uint8 * 4).from_buffer_copy(b"soak")
    O.orc_ed448_sign_batch(_p(want), _p(sk), _p(pk), _p(msg), 48, 0, ctx, 4, N, _gen.NTHREADS)
    assert (sig == want).all()
    rng = np.random.default_rng(5)
    bad = rng.random(N) < 0.25
    sig2 = sig.copy()
    sig2[bad, rng.integers(0, 114, bad.sum())] ^= (1 << rng.integers(0, 8, bad.sum())).astype(np.uint8)
    st = ga.ed448_verify_batch(sig2, pk, msgs, context=b"soak")
    want_st = np.empty(N, np.int32)
    O.orc_ed448_verify_batch(_p(want_st), _p(sig2), _p(pk), _p(msg), 48, 0, ctx, 4, N, _gen.NTHREADS)
    assert (st == want_st).all() and (st[~bad] == -1).all()


@pytest.mark.parametrize("keys", ["combs", "wide combs", "widest combs", "pooled tables"])
def test_soak_verify_repeated_keys(ga, O, keys):
    """The verification kernels for keys that repeat in a batch (kernels_verify.hip: a comb per key and R not decoded /
    a pooled window table per key): 32 768 signatures of 64 keys, a bit flipped anywhere in a quarter of the
    signatures, in an eighth of the keys used and in a few messages, every lane against the oracle."""
    nk = 64
    sk = np.frombuffer(_gen.stream(SEED + b"soak/rk-sk", 57 * nk), np.uint8).reshape(nk, 57).copy()
    pk_k = ga.ed448_derive_public_key_batch(sk)
    rng = np.random.default_rng(11)
    key_of = rng.integers(0, nk, N)
    msg = np.frombuffer(_gen.stream(SEED + b"soak/rk-msg", 40 * N), np.uint8).reshape(N, 40).copy()
    sig = ga.ed448_sign_batch(sk[key_of], pk_k[key_of], [m.tobytes() for m in msg], context=b"rk")
    pk = pk_k[key_of].copy()
    bad = rng.random(N) < 0.25
    sig[bad, rng.integers(0, 114, bad.sum())] ^= (1 << rng.integers(0, 8, bad.sum())).astype(np.uint8)
    badk = rng.random(N) < 0.125
    pk[badk, rng.integers(0, 57, badk.sum())] ^= (1 << rng.integers(0, 8, badk.sum())).astype(np.uint8)
    badm = rng.random(N) < 0.03
    msg[badm, 7] ^= 0x10
    try:
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT, 4097)
        ga.set_verify_key_combs(ga.KEY_COMBS_DEFAULT if "combs" in keys else 0, 2)
        ga.set_verify_key_combs_wide(1 if keys == "wide combs" else 0)
        ga.set_verify_key_combs_xwide(1 if keys == "widest combs" else 0)
        st = ga.ed448_verify_batch(sig, pk, [m.tobytes() for m in msg], context=b"rk")
        distinct, pooled, combed, teeth = ga.last_verify_key_counts(teeth=True)
        assert distinct > nk and (combed if "combs" in keys else pooled) == distinct and teeth == {"combs": 7, "wide combs": 8, "widest combs": 9}.get(keys, 0)
    finally:
        ga.set_verify_key_pool()
        ga.set_verify_key_combs()
        ga.set_verify_key_combs_wide()
        ga.set_verify_key_combs_xwide()
    want = np.empty(N, np.int32)
    ctx = (C.c_uint8 * 2).from_buffer_copy(b"rk")
    O.orc_ed448_verify_batch(_p(want), _p(sig), _p(pk), _p(msg), 40, 0, ctx, 2, N, _gen.NTHREADS)
    assert (st == want).all() and (want == -1).sum() > N // 2 and (want == 0).sum() > N // 4


def test_soak_x448_and_elligator(ga, O):
    from _libs import Point
    n = N // 4
    sc = np.frombuffer(_gen.stream(SEED + b"soak/x448-s", 56 * n), np.uint8).reshape(n, 56).copy()
    pub, _ = ga.x448_batch(sc)
    peer = np.roll(pub, 1, axis=0)
    got, st = ga.x448_batch(sc, peer)
    for i in range(0, n, 7):
        w = (C.c_uint8 * 56)()
        assert O.orc_x448(w, _p(peer[i]), _p(sc[i])) == st[i] and bytes(w) == got[i].tobytes()
    # DH symmetry over the whole batch: x448(a_i, pub_{i-1}) == x448(a_{i-1}, pub_i)
    other, _ = ga.x448_batch(np.roll(sc, 1, axis=0), pub)
    assert (got == other).all()
    h = np.frombuffer(_gen.stream(SEED + b"soak/elligator", 112 * n), np.uint8).reshape(n, 112).copy()
    pts = ga.point_from_hash_batch(h, uniform=True)
    w = np.empty((n, 32), np.uint64)
    for i in range(0, n, 5):
        O.orc_point_from_hash_uniform(C.cast(_p(w[i]), C.POINTER(Point)), _p(h[i]))
    sel = np.arange(0, n, 5)
    assert (enc(ga, pts[sel]) == _gen.oracle_encode(w[sel])).all()


def test_soak_wave_path(ga, O, table_mode):
    """The one-operation-per-wave kernels at their largest batch: 8192 random variable-base multiplications
    and 4096 verifications (a quarter corrupted) against the oracle."""
    if table_mode == "lane_kernels_only":
        pytest.skip("this test is about the wave path")
    n = ga.get_wave_batch_max()
    assert n == 8192
    k = _gen.stream_scalars(n, SEED + b"soak/wave/base")
    s = _gen.stream_scalars(n, SEED + b"soak/wave/scalar")
    bases = _gen.oracle_fixed(O, k)
    got = ga.point_scalarmul_batch(bases, s)
    assert (enc(ga, got) == _gen.oracle_encode(_gen.oracle_varbase(O, bases, s))).all()
    m = n // 2
    sigs, pks, msgs = _gen.signatures(O, m, msglen=33, seed=SEED + b"soak/wave/sig", nkeys=97, context=b"w")
    rng = np.random.default_rng(17)
    bad = rng.random(m) < 0.25
    sigs[bad, rng.integers(0, 114, bad.sum())] ^= (1 << rng.integers(0, 8, bad.sum())).astype(np.uint8)
    st = ga.ed448_verify_batch(sigs, pks, msgs, context=b"w")
    want = np.empty(m, np.int32)
    msg_arr = np.frombuffer(b"".join(msgs), np.uint8).reshape(m, 33).copy()
    ctx = (C.c_uint8 * 1).from_buffer_copy(b"w")
    O.orc_ed448_verify_batch(_p(want), _p(sigs), _p(pks), _p(msg_arr), 33, 0, ctx, 1, m, _gen.NTHREADS)
    assert (st == want).all() and (st[~bad] == -1).all() and (st == 0).sum() >= bad.sum() - 4
    # the codecs and the hash-to-curve map, one operation per wave (at most 1 024 per call)
    from _libs import Point
    c = 1024
    pts = got[:c]
    e56, e57 = ga.point_encode_batch(pts), ga.point_encode_like_eddsa_batch(pts)
    assert (e56 == _gen.oracle_encode(pts)).all()
    d56, st56 = ga.point_decode_batch(e56)
    assert (st56 == -1).sum() >= c - 2 and (ga.point_encode_batch(d56)[st56 == -1] == e56[st56 == -1]).all()
    d57, st57 = ga.point_decode_like_eddsa_batch(e57)
    four = ga.point_scalarmul_batch(pts, _gen.scalars_from_ints([4] * c))
    ok57 = st57 == -1
    assert ok57.sum() >= c - 2 and (ga.point_encode_batch(d57)[ok57] == ga.point_encode_batch(four)[ok57]).all()
    for i in range(0, c, 41):
        out = (C.c_uint8 * 57)()
        O.orc_point_encode_like_eddsa(out, C.cast(_p(np.ascontiguousarray(pts[i])), C.POINTER(Point)))
        assert bytes(out) == e57[i].tobytes()
    h = np.frombuffer(_gen.stream(SEED + b"soak/wave/elligator", 112 * c), np.uint8).reshape(c, 112).copy()
    for uniform in (False, True):
        hp = ga.point_from_hash_batch(h if uniform else h[:, :56].copy(), uniform=uniform)
        w = np.empty((c, 32), np.uint64)
        sel = np.arange(0, c, 13)
        for i in sel:
            if uniform:
                O.orc_point_from_hash_uniform(C.cast(_p(w[i]), C.POINTER(Point)), _p(h[i]))
            else:
                O.orc_point_from_hash_nonuniform(C.cast(_p(w[i]), C.POINTER(Point)), _p(np.ascontiguousarray(h[i, :56])))
        assert (enc(ga, hp[sel]) == _gen.oracle_encode(w[sel])).all()


def test_soak_logs_belong_to_this_toolchain(ga):
    """Parity evidence is tied to the toolchain that produced the code: round 5 found a block the compiler got wrong next to
    the product (docs/history/r05.md H; tools/probes/miscompile_r05_repro.py asks a toolchain whether it still does).  The
    newest extended-soak log under profiles/ that carries a stamp (tools/soak.sh writes `toolchain:` and `library_sha256:`)
    must have been taken with the compiler that built the library under test -- after a toolchain update the soaks are to
    be taken again before their figures are quoted."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stamped = []
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*", "soak_*.txt"))):
        m = re.search(r"^toolchain: (.*)$", open(f).read(), re.M)
        if m:
            stamped.append((f, m.group(1).strip()))
    assert stamped, "no stamped soak log under profiles/ (run tools/soak.sh on the GPU box and commit its summary)"
    newest = max(stamped, key=lambda t: (int(re.search(r"profiles/r(\d+)/", t[0]).group(1)), t[0]))
    assert newest[1] == ga.build_info()["toolchain"], (newest, ga.build_info()["toolchain"])
