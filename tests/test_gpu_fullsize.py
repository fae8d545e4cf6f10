"""BASELINE configs 3 and 4 at their full size (2^20 lanes) through size-independent properties, plus a
sample of lanes bit-exact against the oracle.  (Config 2 at full size: test_full_size_linearity and
test_golden_f6_full_batch_digest in test_gpu_parity.py.)"""
import ctypes as C

import numpy as np
import pytest

import _gen

pytestmark = pytest.mark.gpu
N = 1 << 20


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_full_size_fixed_base_three_ways(ga, O):
    """s*B from the base-point window table, from the LDS comb with a caller table, and from the
    variable-base ladder must be the same group element in every lane."""
    import torch
    s = _gen.stream_scalars(N, b"full/fixed")
    ds = torch.from_numpy(s.view(np.int64)).cuda()
    tab = torch.from_numpy(np.ascontiguousarray(ga.precomputed_base()).view(np.uint8)).cuda()
    base = torch.from_numpy(np.repeat(ga.point_base().reshape(1, 32), N, axis=0).view(np.int64)).cuda()
    a, b, c = (torch.empty((N, 32), dtype=torch.int64, device="cuda") for _ in range(3))
    ga.dev("precomputed_scalarmul", a.data_ptr(), None, ds.data_ptr(), N, None)
    ga.dev("precomputed_scalarmul", b.data_ptr(), tab.data_ptr(), ds.data_ptr(), N, None)
    ga.dev("point_scalarmul", c.data_ptr(), base.data_ptr(), ds.data_ptr(), N, None)
    st = torch.empty(N, dtype=torch.int32, device="cuda")
    for other in (b, c):
        ga.dev("point_pred", st.data_ptr(), a.data_ptr(), other.data_ptr(), 0, N, None)
        assert int((st == -1).sum()) == N
    ga.dev("point_pred", st.data_ptr(), a.data_ptr(), None, 1, N, None)     # on the curve
    assert int((st == -1).sum()) == N
    idx = np.random.default_rng(11).integers(0, N, 512)
    got = ga.point_encode_batch(a.cpu().numpy().view(np.uint64)[idx])
    assert (got == _gen.oracle_encode(_gen.oracle_fixed(O, s[idx]))).all()


def test_large_batches_on_a_callers_table_are_recombed(ga, O):
    """From 2^18 operations on, goldilocks_448_precomputed_scalarmul re-combs the caller's 5 x 5 x 18 table to the
    4 x 7 x 16 comb of twice its base point (k_recomb_big) and halves the scalars: the same group elements as the
    reference comb (a batch just below the threshold), as the variable-base ladder on the table's point, and as
    the oracle; edge scalars first."""
    import torch
    from _libs import Q
    n = (1 << 18) + 77
    point = _gen.oracle_fixed(O, _gen.scalars_from_ints([0x1234567 ** 9 % Q]))[0]
    tab = ga.precompute(point)
    edge = [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 2**445, 2**16, 2**16 - 1, 2**432, 2**433 - 1]
    s = _gen.stream_scalars(n, b"full/recomb")
    s[:len(edge)] = _gen.scalars_from_ints(edge)
    ds = torch.from_numpy(s.view(np.int64)).cuda()
    dtab = torch.from_numpy(np.ascontiguousarray(tab).view(np.uint8)).cuda()
    base = torch.from_numpy(np.repeat(point.reshape(1, 32), n, axis=0).view(np.int64)).cuda()
    a, b, c = (torch.empty((n, 32), dtype=torch.int64, device="cuda") for _ in range(3))
    ga.dev("precomputed_scalarmul", a.data_ptr(), dtab.data_ptr(), ds.data_ptr(), n, None)              # re-combed
    m = (1 << 18) - 1
    ga.dev("precomputed_scalarmul", b.data_ptr(), dtab.data_ptr(), ds.data_ptr(), m, None)              # the reference comb
    ga.dev("point_scalarmul", c.data_ptr(), base.data_ptr(), ds.data_ptr(), n, None)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", st.data_ptr(), a.data_ptr(), c.data_ptr(), 0, n, None)
    assert int((st == -1).sum()) == n
    ga.dev("point_pred", st.data_ptr(), a.data_ptr(), b.data_ptr(), 0, m, None)
    assert int((st[:m] == -1).sum()) == m
    ga.dev("point_pred", st.data_ptr(), a.data_ptr(), None, 1, n, None)     # on the curve
    assert int((st == -1).sum()) == n
    k = 300
    got = ga.point_encode_batch(a[:k].cpu().numpy().view(np.uint64))
    want = _gen.oracle_encode(_gen.oracle_varbase(O, np.repeat(point.reshape(1, 32), k, axis=0), s[:k]))
    assert (got == want).all()


def test_index_independent_sign_over_several_rounds_of_one_launch(ga, O):
    """Key derivation and signing in the library's default mode through the device entry points, 400 017
    operations in ONE launch: every lane handles four operations, and between them the block re-stages the base
    point's comb over the LDS region its SHAKE blocks were hashed in (RestagedCombBig).  Every signature must
    verify; a sample is byte-exact against the oracle (RFC 8032 signing is deterministic)."""
    import ctypes as C
    import torch
    assert ga.get_table_access() == ga.TABLES_INDEX_INDEPENDENT
    n = 400017
    sk_h = np.frombuffer(_gen.stream(b"full/ct-sign/sk", 57 * n), np.uint8).reshape(n, 57).copy()
    msg_h = np.frombuffer(_gen.stream(b"full/ct-sign/msg", 32 * n), np.uint8).reshape(n, 32).copy()
    sk, msg = torch.from_numpy(sk_h).cuda(), torch.from_numpy(msg_h).cuda()
    pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), n, None)
    ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    assert int((st == -1).sum()) == n
    idx = np.concatenate([np.arange(4), np.random.default_rng(19).integers(0, n, 120), np.arange(n - 4, n)])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sub_sk, sub_msg = np.ascontiguousarray(sk_h[idx]), np.ascontiguousarray(msg_h[idx])
    want_pk = np.empty((len(idx), 57), np.uint8)
    want_sig = np.empty((len(idx), 114), np.uint8)
    O.orc_ed448_derive_public_key_batch(p(want_pk), p(sub_sk), len(idx), _gen.NTHREADS)
    O.orc_ed448_sign_batch(p(want_sig), p(sub_sk), p(want_pk), p(sub_msg), 32, 0, None, 0, len(idx), _gen.NTHREADS)
    assert (pk.cpu().numpy()[idx] == want_pk).all() and (sig.cpu().numpy()[idx] == want_sig).all()


def test_large_eddsa_encode_batches_share_their_inversions(ga, O):
    """goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa over 2^19 + 5 points: from two residencies on, the
    points a lane handles share one field inversion (k_point_encode_eddsa_shared).  Same 57 bytes as the
    one-inversion-per-point kernel (a batch below the threshold), and as the oracle on a sample; the identity and
    the points of small order are in the batch."""
    import ctypes as C
    import torch
    from _libs import Point
    n = (1 << 19) + 5
    k = _gen.stream_scalars(n, b"full/eddsa-enc")
    k[:3] = _gen.scalars_from_ints([0, 1, 2])
    dk = torch.from_numpy(k.view(np.int64)).cuda()
    pts = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    ga.dev("precomputed_scalarmul", pts.data_ptr(), None, dk.data_ptr(), n, None)
    a = torch.zeros((n, 57), dtype=torch.uint8, device="cuda")
    b = torch.zeros((n, 57), dtype=torch.uint8, device="cuda")
    ga.dev("point_encode_eddsa", a.data_ptr(), pts.data_ptr(), n, None)                 # shared inversions
    m = 100000
    for lo in range(0, n, m):                                                            # one inversion per point
        cnt = min(m, n - lo)
        ga.dev("point_encode_eddsa", b.data_ptr() + 57 * lo, pts.data_ptr() + 256 * lo, cnt, None)
    assert bool((a == b).all())
    idx = np.concatenate([np.arange(6), np.random.default_rng(23).integers(0, n, 100)])
    ph = pts.cpu().numpy().view(np.uint64)
    got = a.cpu().numpy()
    for i in idx:
        out = (C.c_uint8 * 57)()
        O.orc_point_encode_like_eddsa(out, C.cast(ph[i].ctypes.data_as(C.c_void_p), C.POINTER(Point)))
        assert bytes(out) == got[i].tobytes(), int(i)


def test_full_size_sign_verify_round_trip(ga, O):
    """derive -> sign -> verify on 2^20 independent keys: every signature verifies, exactly the lanes
    whose signature, key or message was corrupted are rejected; a sample is bit-exact vs the oracle."""
    sk = np.frombuffer(_gen.stream(b"full/sk", 57 * N), np.uint8).reshape(N, 57).copy()
    msg = np.frombuffer(_gen.stream(b"full/msg", 32 * N), np.uint8).reshape(N, 32).copy()
    pk = ga.ed448_derive_public_key_batch(sk)
    msgs = [m.tobytes() for m in msg]
    sig = ga.ed448_sign_batch(sk, pk, msgs)
    st = ga.ed448_verify_batch(sig, pk, msgs)
    assert (st == -1).all()
    rng = np.random.default_rng(13)
    which = rng.integers(0, 4, N)                       # 0: untouched, 1: sig, 2: pk, 3: msg
    bad = rng.random(N) < 0.01
    sig2, pk2, msg2 = sig.copy(), pk.copy(), msg.copy()
    m = bad & (which == 1)
    sig2[m, rng.integers(0, 113, m.sum())] ^= (1 << rng.integers(0, 8, m.sum())).astype(np.uint8)
    m = bad & (which == 2)
    pk2[m, rng.integers(0, 56, m.sum())] ^= (1 << rng.integers(0, 8, m.sum())).astype(np.uint8)
    m = bad & (which == 3)
    msg2[m, rng.integers(0, 32, m.sum())] ^= (1 << rng.integers(0, 8, m.sum())).astype(np.uint8)
    touched = bad & (which != 0)
    st2 = ga.ed448_verify_batch(sig2, pk2, [r.tobytes() for r in msg2])
    assert (st2[~touched] == -1).all() and (st2[touched] == 0).all() and touched.sum() > 5000
    idx = np.concatenate([rng.integers(0, N, 256), np.flatnonzero(touched)[:256]])
    k = len(idx)
    want_pk, want_sig, want_st = np.empty((k, 57), np.uint8), np.empty((k, 114), np.uint8), np.empty(k, np.int32)
    ski, msgi = np.ascontiguousarray(sk[idx]), np.ascontiguousarray(msg[idx])
    O.orc_ed448_derive_public_key_batch(_p(want_pk), _p(ski), k, _gen.NTHREADS)
    O.orc_ed448_sign_batch(_p(want_sig), _p(ski), _p(want_pk), _p(msgi), 32, 0, None, 0, k, _gen.NTHREADS)
    assert (pk[idx] == want_pk).all() and (sig[idx] == want_sig).all()
    s2i, p2i, m2i = (np.ascontiguousarray(x[idx]) for x in (sig2, pk2, msg2))
    O.orc_ed448_verify_batch(_p(want_st), _p(s2i), _p(p2i), _p(m2i), 32, 0, None, 0, k, _gen.NTHREADS)
    assert (st2[idx] == want_st).all()


def test_host_array_pipeline_matches_fixture_and_ragged_tail(ga, O):
    """The host-array entry point is software-pipelined for large n: the 2^20 benchmark batch through it
    must hash to the reference's digest (golden F6), and a batch with a ragged last chunk must agree
    lane for lane with the single-launch device path."""
    import hashlib
    import json
    import os
    import torch
    dig = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                      "f6_bench_digest.json")))["digest_shake256_32"]
    k = _gen.stream_scalars(N, b"bench_varbase_v1/0/base")
    s = _gen.stream_scalars(N, b"bench_varbase_v1/0/scalar")
    bases = ga.precomputed_scalarmul_batch(k)
    out = ga.point_scalarmul_batch(bases, s)                      # 8 chunks of one chip residency
    assert hashlib.shake_256(ga.point_encode_batch(out).tobytes()).hexdigest(32) == dig["20"]
    n = 2 * 131072 + 777                                          # pipelined, last chunk ragged
    part = ga.point_scalarmul_batch(bases[:n], s[:n])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    db, dsc = d(bases[:n]), d(s[:n])
    dout = torch.empty_like(db)
    ga.dev("point_scalarmul", dout.data_ptr(), db.data_ptr(), dsc.data_ptr(), n, None)
    torch.cuda.synchronize()
    # same group elements (raw limbs are a projective representation: the 777-operation tail of the
    # pipelined call runs one operation per wave, the single launch one per lane)
    enc = ga.point_encode_batch
    assert (enc(dout.cpu().numpy().view(np.uint64)) == enc(part)).all() and (enc(part) == enc(out[:n])).all()
    # the schedule is a first and a last chunk of one residency and equal chunks of at most six between them: sizes
    # with no middle, a middle of one operation's worth less than a residency, one and two middle chunks, ragged ones
    want = enc(out)
    for m in (2 * 131072 + 1, 3 * 131072 - 1, 3 * 131072, 4 * 131072 + 9, 7 * 131072 + 131071, N - 1):
        assert (enc(ga.point_scalarmul_batch(bases[:m], s[:m])) == want[:m]).all(), m


@pytest.mark.parametrize("keys", ["combs", "pooled", "distinct"])
def test_host_array_verification_pipeline_in_chunks(ga, O, keys):
    """goldilocks_ed448_verify_batch from host arrays, 2^19 signatures or more: the keys' preparation once for the batch,
    the signatures chunk by chunk (2^18) through the first pass of the key-comb verification, ONE shared inversion per
    lane over the chunks of a group (kernels_verify.hip k_ed448_verify_keycomb_finish), or chunk by chunk through
    k_ed448_verify.  2^20 + 777 signatures of 2^10 keys are two groups, the second a ragged tail; the other key
    distributions run 2^19 + 5.  Verdicts: exactly the corrupted lanes fail, lane for lane what the single-launch device
    entry point says, and a sample (rejects among them) what the oracle says (src/eddsa.c:253-306)."""
    import torch
    n = (1 << 20) + 777 if keys == "combs" else (1 << 19) + 5
    nk = {"combs": 1 << 10, "pooled": n // 4, "distinct": n}[keys]
    sk = np.frombuffer(_gen.stream(b"pipe/sk/" + keys.encode(), 57 * nk), np.uint8).reshape(nk, 57)
    which = (np.arange(n) * 2654435761 % nk) if keys != "distinct" else np.arange(n)      # keys in scattered order
    sk_n = np.ascontiguousarray(sk[which])
    msg = np.frombuffer(_gen.stream(b"pipe/msg", 24 * 4096), np.uint8).reshape(4096, 24)[np.arange(n) % 4096].copy()
    msg[:, :4] = np.arange(n, dtype=np.uint32).view(np.uint8).reshape(n, 4)                  # every message distinct
    pk = ga.ed448_derive_public_key_batch(sk_n)
    msgs = [m.tobytes() for m in msg]
    sig = ga.ed448_sign_batch(sk_n, pk, msgs)
    idx = np.arange(n)
    bad = (idx % 97) == 5
    kind = (idx // 97) % 4                                  # corrupt S, R, the key or the message, by turns
    sig[bad & (kind == 0), 60] ^= 1
    sig[bad & (kind == 1), 5] ^= 0x20
    pk[bad & (kind == 2), 9] ^= 4
    msg[bad & (kind == 3), 23] ^= 0x80
    msgs = [m.tobytes() for m in msg]
    st = ga.ed448_verify_batch(sig, pk, msgs)
    assert ((st == -1) | (st == 0)).all()
    # (a corrupted key is a key of its own: it may fail to decode or verify, never accept)
    assert ((st == -1) == ~bad).all(), (int((st == -1).sum()), int((~bad).sum()))
    counts = ga.last_verify_key_counts()
    if keys == "combs":
        assert counts[2] > 0 and counts[1] == 0, counts     # the batch's keys got combs (the corrupted ones too)
    elif keys == "pooled":
        assert counts[1] > 0 and counts[2] == 0, counts
    else:
        assert counts[1] == 0 and counts[2] == 0, counts
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dst = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    dsig, dpk, dmsg = d(sig), d(pk), d(msg)
    ga.dev("ed448_verify", dst.data_ptr(), dsig.data_ptr(), dpk.data_ptr(), dmsg.data_ptr(), None, 24, 0, None, 0, n, None)
    torch.cuda.synchronize()
    assert (dst.cpu().numpy() == st).all()
    rng = np.random.default_rng(len(keys))
    pick = np.unique(np.concatenate([rng.integers(0, n, 150), 97 * rng.integers(0, n // 97, 60) + 5, np.arange(n - 40, n)]))
    want = _gen.oracle_verify(O, sig[pick], pk[pick], [msgs[i] for i in pick])
    assert (st[pick] == want).all() and (want == 0).sum() >= 50


@pytest.mark.parametrize("n", [131071, 131072, 131073, 3 * 131072 + 5, (1 << 19) - 1, 1 << 19])
def test_verification_sizes_around_the_launch_and_pipeline_thresholds(ga, O, n):
    """The phased verification at the sizes where its plan changes: one lane short of a full grid, exactly a full grid (the
    first size with S*B computed ahead), one more, a ragged multi-round batch, and the host-array entry point one signature
    below and exactly at the size from which it pipelines (two chunks).  64 keys (combs), every 53rd signature corrupted in
    turn in S, R, key or message: the accept set must be exactly the untouched lanes, through the device entry point and
    through the host arrays alike."""
    import torch
    nk = 64
    sk = np.frombuffer(_gen.stream(b"thr/sk", 57 * nk), np.uint8).reshape(nk, 57)
    which = np.arange(n) * 40503 % nk
    sk_n = np.ascontiguousarray(sk[which])
    msg = np.zeros((n, 16), np.uint8)
    msg[:, :4] = np.arange(n, dtype=np.uint32).view(np.uint8).reshape(n, 4)
    pk = ga.ed448_derive_public_key_batch(sk_n)
    sig = ga.ed448_sign_batch(sk_n, pk, [m.tobytes() for m in msg])
    idx = np.arange(n)
    bad = (idx % 53) == 7
    kind = (idx // 53) % 4
    sig[bad & (kind == 0), 70] ^= 2
    sig[bad & (kind == 1), 11] ^= 0x10
    pk[bad & (kind == 2), 3] ^= 1
    msg[bad & (kind == 3), 15] ^= 0x40
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dst = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    dsig, dpk, dmsg = d(sig), d(pk), d(msg)
    ga.dev("ed448_verify", dst.data_ptr(), dsig.data_ptr(), dpk.data_ptr(), dmsg.data_ptr(), None, 16, 0, None, 0, n, None)
    torch.cuda.synchronize()
    st = dst.cpu().numpy()
    assert ((st == -1) == ~bad).all() and ((st == -1) | (st == 0)).all()
    assert ga.last_verify_key_counts()[2] > 0                      # combs
    st2 = ga.ed448_verify_batch(sig, pk, [m.tobytes() for m in msg])
    assert (st2 == st).all()


def test_sharded_host_batches_match_single_device(ga, O):
    """goldilocks_amd_use_devices: the contiguous-slice sharding of the host-array batches (one host
    thread per listed device).  A 1-GPU box lists device 0 three times, so the shards run one after
    the other through the same code path; results must equal the unsharded call lane for lane."""
    n = 3 * 4099 + 2                                     # ragged slices: 4099|4100|4100 -ish
    k = _gen.stream_scalars(n, b"shard/base")
    s = _gen.stream_scalars(n, b"shard/scalar")
    sigs, pks, msgs = _gen.signatures(O, 700, msglen=21, seed=b"shard-sig")
    sigs[5, 3] ^= 4
    sigs[699, 60] ^= 1
    want_fixed = ga.precomputed_scalarmul_batch(k)
    want_var = ga.point_scalarmul_batch(want_fixed, s)
    want_st = ga.ed448_verify_batch(sigs, pks, msgs)
    ga.use_devices([0, 0, 0])
    try:
        got_fixed = ga.precomputed_scalarmul_batch(k)
        got_var = ga.point_scalarmul_batch(got_fixed, s)
        got_st = ga.ed448_verify_batch(sigs, pks, msgs)
        with pytest.raises(ga.GoldilocksAmdError):
            ga.use_devices([0, 99])                      # not a visible device
        # a single-operation drop-in call still works with sharding configured
        assert (ga.point_encode_batch(ga.point_scalarmul(got_fixed[0], s[0]).reshape(1, 32)) ==
                ga.point_encode_batch(want_var[:1])).all()
    finally:
        ga.use_devices(None)
    # the devices (and the table access) of ONE call, the process-wide list untouched (the *_batch_ex entry points)
    call_fixed = ga.precomputed_scalarmul_batch(k, devices=[0, 0])
    call_var = ga.point_scalarmul_batch(call_fixed, s, flags=ga.CALL_TABLES_FAST, devices=[0, 0, 0, 0])
    call_st = ga.ed448_verify_batch(sigs, pks, msgs, devices=[0, 0, 0])
    with pytest.raises(ga.GoldilocksAmdError):
        ga.point_scalarmul_batch(call_fixed, s, devices=[0, 99])
    with pytest.raises(ga.GoldilocksAmdError):
        ga.point_scalarmul_batch(call_fixed, s, flags=8)
    # group elements, not raw limbs: a shard of 4 099 operations runs one operation per wave, the
    # unsharded 12 299 one per lane -- different projective representatives of the same points
    enc = ga.point_encode_batch
    assert (enc(call_fixed) == enc(want_fixed)).all() and (enc(call_var) == enc(want_var)).all() and (call_st == want_st).all()
    assert (enc(got_fixed) == enc(want_fixed)).all() and (enc(got_var) == enc(want_var)).all()
    assert (got_st == want_st).all() and got_st[5] == 0 and got_st[699] == 0 and (got_st == -1).sum() == 698
    assert (ga.point_encode_batch(got_var[:64]) == _gen.oracle_encode(_gen.oracle_varbase(O, want_fixed[:64], s[:64]))).all()


def test_concurrent_host_threads_on_one_device(ga, O):
    """The reference's functions are reentrant (SURVEY 8b "Threading"); here calls for one device
    serialize on that device's lock.  Four host threads issue different batches at once (ctypes drops
    the GIL around the calls) and every result must be the one a lone caller gets."""
    import threading
    n = 3000
    k = [_gen.stream_scalars(n, b"thr/base/%d" % t) for t in range(4)]
    s = [_gen.stream_scalars(n, b"thr/scalar/%d" % t) for t in range(4)]
    want = []
    for t in range(4):
        b = ga.precomputed_scalarmul_batch(k[t])
        want.append((b, ga.point_scalarmul_batch(b, s[t])))
    got = [None] * 4
    errs = []

    def work(t):
        try:
            for _ in range(3):
                b = ga.precomputed_scalarmul_batch(k[t])
                got[t] = (b, ga.point_scalarmul_batch(b, s[t]))
        except Exception as e:                           # noqa: BLE001 - reported below
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(4):
        assert (got[t][0] == want[t][0]).all() and (got[t][1] == want[t][1]).all()    # same kernels, same limbs


def test_index_independent_tables_match_fast_tables(ga, O):
    """Derive / sign / X448 keygen / base-point scalarmul through the LDS comb + wavefront-shuffle gather
    (the default, index-independent) give the same bytes as the opt-in window-table kernels
    (GOLDILOCKS_AMD_TABLES_FAST), on ragged batch sizes (idle lanes of the last wave redo the last operation)."""
    for n in (1, 63, 3001):
        sk = np.frombuffer(_gen.stream(b"ct/sk/%d" % n, 57 * n), np.uint8).reshape(n, 57).copy()
        xs = np.frombuffer(_gen.stream(b"ct/x448/%d" % n, 56 * n), np.uint8).reshape(n, 56).copy()
        msgs = [_gen.stream(b"ct/msg/%d/%d" % (n, i), i % 200) for i in range(n)]     # ragged lengths
        k = _gen.stream_scalars(n, b"ct/scalar/%d" % n)
        ga.set_table_access(ga.TABLES_FAST)
        try:
            pk0 = ga.ed448_derive_public_key_batch(sk)
            sig0 = ga.ed448_sign_batch(sk, pk0, msgs, context=b"ct")
            x0, _ = ga.x448_batch(xs)
            b0 = ga.precomputed_scalarmul_batch(k)
            with pytest.raises(ga.GoldilocksAmdError):
                ga.set_table_access(7)
        finally:
            ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
        assert ga.get_table_access() == ga.TABLES_INDEX_INDEPENDENT
        pk1 = ga.ed448_derive_public_key_batch(sk)
        sig1 = ga.ed448_sign_batch(sk, pk0, msgs, context=b"ct")
        x1, _ = ga.x448_batch(xs)
        b1 = ga.precomputed_scalarmul_batch(k)
        assert (pk0 == pk1).all() and (sig0 == sig1).all() and (x0 == x1).all()
        assert (ga.point_encode_batch(b0) == ga.point_encode_batch(b1)).all()
        assert (ga.ed448_verify_batch(sig1, pk1, msgs, context=b"ct") == -1).all()
        m = min(n, 64)
        want_pk = np.empty((m, 57), np.uint8)
        O.orc_ed448_derive_public_key_batch(_p(want_pk), _p(np.ascontiguousarray(sk[:m])), m, _gen.NTHREADS)
        assert (pk1[:m] == want_pk).all()


def test_shared_inversion_chains_with_zero_denominators(ga, O):
    """derive / sign / X448 share one field inversion between the operations a lane handles back to
    back (Montgomery's trick, fixed_bodies.hpp).  With more operations than resident lanes, put
    low-order X448 inputs (result 0, denominator 0) at several positions of the same lane's chain and
    compare those lanes, their neighbours and a random sample with the oracle one by one."""
    lanes = ga.device_info()["compute_units"] * 2 * 256
    n = 3 * lanes + 5
    sc = np.frombuffer(_gen.stream(b"inv/x448-s", 56 * n), np.uint8).reshape(n, 56).copy()
    bs = np.frombuffer(_gen.stream(b"inv/x448-b", 56 * n), np.uint8).reshape(n, 56).copy()
    zero_at = [7, 7 + lanes, 9 + 2 * lanes, 11, 11 + lanes, 11 + 2 * lanes, n - 1]
    for j, i in enumerate(zero_at):
        bs[i] = 0
        if j % 2:
            bs[i, 0] = 1                       # u = 1 is low-order as well
    got, st = ga.x448_batch(sc, bs)
    rng = np.random.default_rng(3)
    check = sorted(set(zero_at + [7 + 2 * lanes, 9, 9 + lanes, 0, 1, lanes, 2 * lanes, 3 * lanes]
                       + list(rng.integers(0, n, 200))))
    w = (C.c_uint8 * 56)()
    for i in check:
        assert O.orc_x448(w, bs[i].ctypes.data, sc[i].ctypes.data) == st[i], i
        assert bytes(w) == got[i].tobytes(), i
    assert all(st[i] == 0 and not got[i].any() for i in zero_at)
    assert (st == -1).sum() == n - len(zero_at)
    # sign with three operations per lane against the oracle on a sample (ragged message lengths)
    m = 2 * lanes + 77
    sk = np.frombuffer(_gen.stream(b"inv/sk", 57 * m), np.uint8).reshape(m, 57).copy()
    pk = ga.ed448_derive_public_key_batch(sk)
    msgs = [bytes([i & 255]) * (i % 70) for i in range(m)]
    sig = ga.ed448_sign_batch(sk, pk, msgs, context=b"chain")
    assert (ga.ed448_verify_batch(sig, pk, msgs, context=b"chain") == -1).all()
    ctx = (C.c_uint8 * 5).from_buffer_copy(b"chain")
    for i in [0, 1, lanes - 1, lanes, lanes + 1, 2 * lanes, m - 1] + list(rng.integers(0, m, 60)):
        wpk, wsig = (C.c_uint8 * 57)(), (C.c_uint8 * 114)()
        O.orc_ed448_derive_public_key(wpk, sk[i].ctypes.data)
        mb = (C.c_uint8 * max(1, len(msgs[i]))).from_buffer_copy(msgs[i] or b"\0")
        O.orc_ed448_sign(wsig, sk[i].ctypes.data, wpk, mb, len(msgs[i]), 0, ctx, 5)
        assert bytes(wpk) == pk[i].tobytes() and bytes(wsig) == sig[i].tobytes(), i


def test_sub_batched_launches_beyond_eight_operations_per_lane(ga, O):
    """More than 8 operations per resident lane: the host splits the batch into several launches that
    reuse the per-operation workspace; results must not depend on where the split falls."""
    lanes = ga.device_info()["compute_units"] * 2 * 256
    n = 8 * lanes + 1234
    sk = np.frombuffer(_gen.stream(b"sub/sk", 57 * n), np.uint8).reshape(n, 57).copy()
    pk = ga.ed448_derive_public_key_batch(sk)
    tail = ga.ed448_derive_public_key_batch(sk[8 * lanes - 50:])          # a different split of the same keys
    assert (pk[8 * lanes - 50:] == tail).all()
    idx = np.concatenate([np.arange(8 * lanes - 3, 8 * lanes + 3), np.array([0, n - 1]),
                          np.random.default_rng(4).integers(0, n, 100)])
    want = np.empty((len(idx), 57), np.uint8)
    O.orc_ed448_derive_public_key_batch(_p(want), _p(np.ascontiguousarray(sk[idx])), len(idx), _gen.NTHREADS)
    assert (pk[idx] == want).all()
    xs = sk[:, :56].copy()
    pub, st = ga.x448_batch(xs)
    assert (st == -1).all()
    w = (C.c_uint8 * 56)()
    for i in [0, 8 * lanes - 1, 8 * lanes, n - 1]:
        O.orc_x448_derive_public_key(w, xs[i].ctypes.data)
        assert bytes(w) == pub[i].tobytes(), i


def test_shutdown_releases_and_next_call_rebuilds(ga, O):
    """goldilocks_amd_shutdown frees the device context; the next call builds it again lazily."""
    k = _gen.stream_scalars(100, b"shutdown")
    before = ga.point_encode_batch(ga.precomputed_scalarmul_batch(k))
    assert ga.device_info()["workspace_bytes"] > 0
    ga.lib().goldilocks_amd_shutdown()
    assert ga.device_info()["workspace_bytes"] == 0            # fresh context, nothing staged yet
    after = ga.point_encode_batch(ga.precomputed_scalarmul_batch(k))
    assert (before == after).all() and (after == _gen.oracle_encode(_gen.oracle_fixed(O, k))).all()


def test_batch_output_may_alias_the_base_array(ga, O):
    """The reference lets outputs alias inputs (point_448.h:295-297); for the batch entry point that means
    scaled == base, also when the batch is large enough to be pipelined in chunks."""
    n = 2 * 131072 + 4321
    k = _gen.stream_scalars(n, b"alias/base")
    s = _gen.stream_scalars(n, b"alias/scalar")
    bases = ga.precomputed_scalarmul_batch(k)
    want = ga.point_scalarmul_batch(bases, s)
    buf = bases.copy()
    rc = ga.lib().goldilocks_448_point_scalarmul_batch(buf.ctypes.data, buf.ctypes.data, s.ctypes.data, n)
    enc = ga.point_encode_batch
    assert rc == 0 and (enc(buf) == enc(want)).all()
    small = bases[:100].copy()                                     # 100 operations: the one-operation-per-wave path
    assert ga.lib().goldilocks_448_point_scalarmul_batch(small.ctypes.data, small.ctypes.data, s.ctypes.data, 100) == 0
    assert (enc(small) == enc(want[:100])).all()


@pytest.mark.parametrize("log2n", [20, 21])
def test_config5_share_of_one_gpu_in_one_launch(ga, O, log2n):
    """BASELINE config 5 at its real per-GPU size: 2^21 verifications (2^24 over 8 GPUs) in ONE
    goldilocks_amd_ed448_verify_dev launch, a tenth of them corrupted -- and 2^20, the launch bench.py times
    (config 4).  The accept / reject set must be exact in every lane (reference: src/eddsa.c:253-306, early
    returns and all), and 256 lanes spread over the batch, rejects among them, are re-verified by the oracle."""
    import torch
    n, nk = 1 << log2n, 1024
    sk = np.frombuffer(_gen.stream(b"c5/sk", 57 * nk), np.uint8).reshape(nk, 57)
    sk_d = torch.from_numpy(np.ascontiguousarray(sk[np.arange(n) % nk])).cuda()
    msg = torch.from_numpy(np.frombuffer(_gen.stream(b"c5/msg", 32 * 4096), np.uint8).reshape(4096, 32)[np.arange(n) % 4096].copy()).cuda()
    idx = torch.arange(n, device="cuda")
    for b in range(3):                                     # every message distinct
        msg[:, b] = ((idx >> (8 * b)) & 0xff).to(torch.uint8)
    pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ga.dev("ed448_derive_public_key", pk.data_ptr(), sk_d.data_ptr(), n, None)
    ga.dev("ed448_sign", sig.data_ptr(), sk_d.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    bad = (idx % 10) == 3
    # corrupt S, R, the key or the message, by turns
    kind = (idx // 10) % 4
    sig[bad & (kind == 0), 60] ^= 1
    sig[bad & (kind == 1), 5] ^= 0x20
    pk[bad & (kind == 2), 9] ^= 4
    msg[bad & (kind == 3), 31] ^= 0x80
    ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    torch.cuda.synchronize()
    assert bool(((st == -1) | (st == 0)).all())
    assert bool(((st == -1) == ~bad).all())
    rng = np.random.default_rng(log2n)
    pick = np.unique(np.concatenate([rng.integers(0, n, 200), 10 * rng.integers(0, n // 10, 56) + 3]))   # 56 corrupted lanes for sure
    pick_d = torch.from_numpy(pick).cuda()
    want = _gen.oracle_verify(O, sig[pick_d].cpu().numpy(), pk[pick_d].cpu().numpy(),
                              [m.tobytes() for m in msg[pick_d].cpu().numpy()])
    assert (st[pick_d].cpu().numpy() == want).all() and (want == 0).sum() >= 50


@pytest.mark.parametrize("log2n,nk,teeth", [(19, 500, 9), (19, 1500, 8), (18, 3000, 7)])
def test_key_combs_built_in_ragged_segments(ga, O, log2n, nk, teeth):
    """k_verify_key_combs cuts a comb's Gray-code walk into segments whose length follows the number of keys
    (kernels_verify.hip key_comb_segment): lengths that do not divide the comb leave a shorter last segment.  Key counts
    that give such lengths for each comb geometry -- 500 keys of 9 teeth: 26 segments of 10 and one of 6 per comb; 1 500
    keys of 8 teeth: 13, last 11; 3 000 keys of 7 teeth: 13, last 12 -- every verdict exact, a sample against the oracle."""
    import torch
    n = 1 << log2n
    sk = torch.from_numpy(np.frombuffer(_gen.stream(b"ragged/sk%d" % nk, 57 * nk), np.uint8).reshape(nk, 57).copy()).cuda()
    idx = torch.arange(n, device="cuda")
    d_sk = sk[idx % nk].contiguous()
    d_msg = torch.from_numpy(np.frombuffer(_gen.stream(b"ragged/msg", 32 * 4096), np.uint8).reshape(4096, 32)[np.arange(n) % 4096].copy()).cuda()
    for b in range(3):
        d_msg[:, b] = ((idx >> (8 * b)) & 0xff).to(torch.uint8)
    d_pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
    d_sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_derive_public_key", d_pk.data_ptr(), d_sk.data_ptr(), n, None)
    ga.dev("ed448_sign", d_sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    bad = (idx % 10) == 7
    kind = (idx // 10) % 3
    d_sig[bad & (kind == 0), 70] ^= 2       # S
    d_sig[bad & (kind == 1), 8] ^= 0x10     # R
    d_msg[bad & (kind == 2), 30] ^= 1       # the message
    st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ga.dev("ed448_verify", st.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    torch.cuda.synchronize()
    assert ga.last_verify_key_counts() == (nk, 0, nk)
    # the geometry this batch gets (k_verify_key_mode) and the segment length that follows from it
    per_key = n // nk
    assert teeth == (9 if nk * 1024 * (2 if nk > 1024 else 1) <= n else 8 if nk * 256 <= n else 7)
    combs, per_comb = nk * (5 if teeth == 9 else 4), 1 << (teeth - 1)
    room = max(1, min(per_comb // 8, 131072 // (2 * combs)))
    assert per_comb % -(-per_comb // room) != 0, "this case is meant to leave a shorter last segment"
    assert bool(((st == -1) == ~bad).all()), per_key
    pick = np.unique(np.concatenate([np.random.default_rng(nk).integers(0, n, 64), 10 * np.random.default_rng(nk + 1).integers(0, n // 10, 32) + 7]))
    pick_d = torch.from_numpy(pick).cuda()
    want = _gen.oracle_verify(O, d_sig[pick_d].cpu().numpy(), d_pk[pick_d].cpu().numpy(), [m.tobytes() for m in d_msg[pick_d].cpu().numpy()])
    assert (st[pick_d].cpu().numpy() == want).all() and (want == 0).sum() >= 30


def test_verification_shares_the_tables_of_repeated_keys(ga, O):
    """For large batches the verification kernel decodes every DISTINCT public key once and builds its window table
    once (goldilocks_amd_set_verify_key_pool; kernels_verify.hip), or -- keys that sign many signatures each -- a
    fixed-base comb per key (goldilocks_amd_set_verify_key_combs).  Verdicts must not depend on it: the same batch
    -- signatures of 37 keys, rejects of every kind, an undecodable key that many signatures share -- with the pool
    off, with the default pool, with a pool too small for the batch's keys (then nothing is pooled) and with a batch
    of all-distinct keys (no pool either: more than half of the signatures bring their own)."""
    import torch
    n, nk = 1 << 17, 37
    sk = np.frombuffer(_gen.stream(b"pool/sk", 57 * nk), np.uint8).reshape(nk, 57)
    pk_k = ga.ed448_derive_public_key_batch(sk)
    key_of = np.random.default_rng(5).integers(0, nk, n)
    msg_h = np.frombuffer(_gen.stream(b"pool/msg", 24 * n), np.uint8).reshape(n, 24).copy()
    sigs = ga.ed448_sign_batch(sk[key_of], pk_k[key_of], [m.tobytes() for m in msg_h])
    pks = pk_k[key_of].copy()
    idx = np.arange(n)
    kind = idx % 16
    sigs[kind == 3, 60] ^= 1                    # S
    sigs[kind == 5, 9] ^= 0x40                  # R
    msg_h[kind == 7, 0] ^= 1                    # message
    own = (idx % 32) == 9
    pks[own, 11] ^= 2                           # a key of its own (4 096 of them), most likely undecodable or wrong
    for b in range(3):
        pks[own, 20 + b] ^= ((idx[own] >> (8 * b)) & 0xff).astype(np.uint8)
    bad_key = pk_k[0].copy(); bad_key[56] = 0x01   # byte 56 neither 0 nor 0x80: the reference rejects the key
    pks[kind == 11] = bad_key                   # ... shared by n/16 signatures
    want_bad = (kind == 3) | (kind == 5) | (kind == 7) | own | (kind == 11)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_sig, d_pk, d_msg = d(sigs), d(pks), d(msg_h)

    def run(d_sig, d_pk, d_msg, m):
        st = torch.full((m,), 7, dtype=torch.int32, device="cuda")
        ga.dev("ed448_verify", st.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 24, 0, None, 0, m, None)
        torch.cuda.synchronize()
        return st.cpu().numpy()
    try:
        got, served = {}, {}
        # (pool keys, min batch, comb keys, min signatures per key): the batch has 37 + 1 + 4 096 distinct keys
        modes = dict(off=(0, 0, 0, 1), tables=(ga.KEY_POOL_DEFAULT, 1 << 16, 0, 1), tiny=(5, 1 << 16, 0, 1),
                     combs=(ga.KEY_POOL_DEFAULT, 1 << 16, 1 << 13, 8), default=(ga.KEY_POOL_DEFAULT, 1 << 16, ga.KEY_COMBS_DEFAULT, ga.KEY_COMBS_MIN_PER_KEY_DEFAULT))
        for name, (keys, min_batch, comb_keys, per_key) in modes.items():
            ga.set_verify_key_pool(keys, min_batch)
            ga.set_verify_key_combs(comb_keys, per_key)
            got[name] = run(d_sig, d_pk, d_msg, n)
            served[name] = ga.last_verify_key_counts()
        nkeys = len({bytes(k) for k in pks})
        assert served == dict(off=(0, 0, 0), tables=(nkeys, nkeys, 0), tiny=(nkeys, 0, 0), combs=(nkeys, 0, nkeys),
                              default=(nkeys, 0, nkeys)), served          # (default at 2^17 signatures: combs from 16 signatures per key; this batch has 31.7)
        # few keys only (the lanes whose key is one of a kind left out): 2^16 signatures of 38 keys -> combs by default
        few = np.flatnonzero(~own)[: 1 << 16]
        d_few = torch.from_numpy(few).cuda()
        ga.set_verify_key_pool()
        ga.set_verify_key_combs()
        st_few = run(d_sig[d_few].contiguous(), d_pk[d_few].contiguous(), d_msg[d_few].contiguous(), len(few))
        assert ga.last_verify_key_counts() == (38, 0, 38)
        ga.set_verify_key_combs(0, 1)
        assert (run(d_sig[d_few].contiguous(), d_pk[d_few].contiguous(), d_msg[d_few].contiguous(), len(few)) == st_few).all()
        assert (st_few == got["off"][few]).all()
        for name, st in got.items():
            assert set(np.unique(st)) <= {-1, 0}, name
            assert ((st == 0) >= want_bad).all(), name           # every corrupted lane is rejected
            assert (st == got["off"]).all(), name                # and the pool changes no verdict
        assert (got["off"] == 0).sum() <= want_bad.sum() and (got["off"][~want_bad] == -1).all()
        pick = np.concatenate([np.arange(64), np.random.default_rng(6).integers(0, n, 192)])
        want = _gen.oracle_verify(O, sigs[pick], pks[pick], [m.tobytes() for m in msg_h[pick]])
        assert (got["default"][pick] == want).all()
        # all-distinct keys: 2^16 signatures of 2^16 keys through the pooled entry point
        m = 1 << 16
        sk2 = np.frombuffer(_gen.stream(b"pool/sk2", 57 * m), np.uint8).reshape(m, 57).copy()
        d_sk2 = d(sk2)
        d_pk2 = torch.empty((m, 57), dtype=torch.uint8, device="cuda")
        d_sig2 = torch.empty((m, 114), dtype=torch.uint8, device="cuda")
        ga.dev("ed448_derive_public_key", d_pk2.data_ptr(), d_sk2.data_ptr(), m, None)
        ga.dev("ed448_sign", d_sig2.data_ptr(), d_sk2.data_ptr(), d_pk2.data_ptr(), d_msg.data_ptr(), None, 24, 0, None, 0, m, None)
        d_sig2[::9, 70] ^= 4
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT, 1 << 15)
        ga.set_verify_key_combs()
        st2 = run(d_sig2, d_pk2, d_msg, m)
        bad2 = (np.arange(m) % 9) == 0
        assert ((st2 == -1) == ~bad2).all()
    finally:
        ga.set_verify_key_pool()
        ga.set_verify_key_combs()


def test_verification_with_combs_for_ten_thousand_keys(ga, O):
    """More keys than a wave each can serve (kernels_verify.hip: beyond 4 096 keys a LANE computes a key's teeth, beyond
    8 192 the counting sort uses plain atomics): 2^18 signatures of 10 000 keys, a tenth corrupted, through the combs (the
    library's choice at 26 signatures per key), against the same batch with every lane for itself and the oracle."""
    import torch
    n, nk = 1 << 18, 10_000
    sk = torch.from_numpy(np.frombuffer(_gen.stream(b"tenk/sk", 57 * nk), np.uint8).reshape(nk, 57).copy()).cuda()
    pk_k = torch.empty((nk, 57), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_derive_public_key", pk_k.data_ptr(), sk.data_ptr(), nk, None)
    key_of = torch.from_numpy(np.random.default_rng(8).integers(0, nk, n)).cuda()
    d_sk, d_pk = sk[key_of].contiguous(), pk_k[key_of].contiguous()
    d_msg = torch.from_numpy(np.frombuffer(_gen.stream(b"tenk/msg", 32 * n), np.uint8).reshape(n, 32).copy()).cuda()
    d_sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_sign", d_sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    bad = torch.arange(n, device="cuda") % 10 == 3
    d_sig[bad, 3] ^= 0x10                                # R
    d_sig[torch.arange(n, device="cuda") % 50 == 7, 80] ^= 1   # S
    bad |= torch.arange(n, device="cuda") % 50 == 7

    def run():
        st = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        ga.dev("ed448_verify", st.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
        torch.cuda.synchronize()
        return st
    try:
        st = run()
        distinct, pooled, combed = ga.last_verify_key_counts()
        assert distinct == combed == len(np.unique(key_of.cpu().numpy())) > 8192 and pooled == 0
        ga.set_verify_key_pool(0, 0)
        assert bool((run() == st).all())
    finally:
        ga.set_verify_key_pool()
    assert bool(((st == -1) == ~bad).all())
    pick = np.random.default_rng(9).integers(0, n, 256)
    pick_d = torch.from_numpy(pick).cuda()
    want = _gen.oracle_verify(O, d_sig[pick_d].cpu().numpy(), d_pk[pick_d].cpu().numpy(), [m.tobytes() for m in d_msg[pick_d].cpu().numpy()])
    assert (st[pick_d].cpu().numpy() == want).all() and (want == 0).sum() >= 10
