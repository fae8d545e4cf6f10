"""Deterministic synthetic inputs and oracle batch helpers shared by tests, smoke() and bench.py.

Inputs follow SURVEY.md section 8(d): a SHAKE256 stream per label; scalars are 72 stream
bytes reduced mod q (mirrors Scalar(rng), reference point_448.hxx:105-108); base points
are k*B for stream scalars k (computed by the library under test and parity-checked
against the oracle, or by the oracle itself); signatures come from the oracle's signer.
"""
import ctypes as C
import hashlib
import os

import numpy as np

Q = 2**446 - 0x8335DC163BB124B65129C96FDE933D8D723A70AADC873D6D54A7BB0D
NTHREADS = max(1, min(os.cpu_count() or 1, 64))


def stream(seed, nbytes):
    return hashlib.shake_256(b"libgoldilocks_amd/" + bytes(seed)).digest(nbytes)


def random_scalars(n, seed=b"scalars"):
    raw = np.frombuffer(stream(seed, 72 * n), dtype=np.uint8).reshape(n, 72)
    out = np.empty((n, 7), dtype=np.uint64)
    for i in range(n):
        v = int.from_bytes(raw[i].tobytes(), "little") % Q
        out[i] = np.frombuffer(v.to_bytes(56, "little"), dtype=np.uint64)
    return out


def stream_scalars(n, label):
    """n scalars below 2^446 straight from the SHAKE256 stream (56 bytes each, top word masked to 62
    bits; the chance of landing in [q, 2^446) is 2^-222).  This is the benchmark input stream
    (labels bench_varbase_v1/<rank> ...), cheap enough for 2^20 and reproducible bit for bit."""
    s = np.frombuffer(stream(label, 56 * n), dtype=np.uint64).reshape(n, 7).copy()
    s[:, 6] &= np.uint64(2**62 - 1)
    return s


def scalars_from_ints(vals):
    out = np.empty((len(vals), 7), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer((v % Q).to_bytes(56, "little"), dtype=np.uint64)
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_fixed(O, scalars, table=None):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = len(scalars)
    out = np.empty((n, 32), dtype=np.uint64)
    tab = O.orc_precomputed_base() if table is None else _p(np.ascontiguousarray(table, dtype=np.uint64))
    O.orc_precomputed_scalarmul_batch(_p(out), C.cast(tab, C.c_void_p), _p(scalars), n, NTHREADS)
    return out


def oracle_varbase(O, bases, scalars):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = len(scalars)
    out = np.empty((n, 32), dtype=np.uint64)
    O.orc_point_scalarmul_batch(_p(out), _p(bases), _p(scalars), n, NTHREADS)
    return out


def oracle_double(O, b1, s1, b2, s2):
    """s1[i] * b1[i] + s2[i] * b2[i] by the oracle's goldilocks_448_point_double_scalarmul, one call per operation"""
    from _libs import Point, Scalar
    b1, s1, b2, s2 = (np.ascontiguousarray(a, dtype=np.uint64) for a in (b1, s1, b2, s2))
    out = np.empty((len(s1), 32), dtype=np.uint64)
    pt = lambda a: C.cast(_p(a), C.POINTER(Point))
    sc = lambda a: C.cast(_p(a), C.POINTER(Scalar))
    for i in range(len(s1)):
        O.orc_point_double_scalarmul(pt(out[i]), pt(b1[i]), sc(s1[i]), pt(b2[i]), sc(s2[i]))
    return out


def oracle_encode(points):
    from _libs import oracle
    O = oracle()
    points = np.ascontiguousarray(points, dtype=np.uint64)
    out = np.empty((len(points), 56), dtype=np.uint8)
    O.orc_point_encode_batch(_p(out), _p(points), len(points), NTHREADS)
    return out


def signatures(O, n, msglen=32, seed=b"sigs", nkeys=None, context=b"", prehashed=False):
    """n valid Ed448 signatures over msglen-byte stream messages from nkeys stream keys."""
    nkeys = nkeys or n
    sk = np.frombuffer(stream(seed + b"/sk", 57 * nkeys), dtype=np.uint8).reshape(nkeys, 57).copy()
    pk = np.empty((nkeys, 57), dtype=np.uint8)
    O.orc_ed448_derive_public_key_batch(_p(pk), _p(sk), nkeys, NTHREADS)
    idx = np.arange(n) % nkeys
    sks, pks = np.ascontiguousarray(sk[idx]), np.ascontiguousarray(pk[idx])
    msgs = np.frombuffer(stream(seed + b"/msg", max(1, msglen * n)), dtype=np.uint8)[:msglen * n].reshape(n, msglen).copy()
    sigs = np.empty((n, 114), dtype=np.uint8)
    ctx = (C.c_uint8 * max(1, len(context))).from_buffer_copy(bytes(context) or b"\0")
    O.orc_ed448_sign_batch(_p(sigs), _p(sks), _p(pks), _p(msgs), msglen, 1 if prehashed else 0, ctx, len(context),
                           n, NTHREADS)
    return sigs, pks, [m.tobytes() for m in msgs]


def oracle_verify(O, sigs, pks, msgs, context=b"", prehashed=False):
    n = len(sigs)
    out = np.empty(n, dtype=np.int32)
    ctx = (C.c_uint8 * max(1, len(context))).from_buffer_copy(bytes(context) or b"\0")
    for i in range(n):
        m = (C.c_uint8 * max(1, len(msgs[i]))).from_buffer_copy(bytes(msgs[i]) or b"\0")
        out[i] = O.orc_ed448_verify(_p(np.ascontiguousarray(sigs[i])), _p(np.ascontiguousarray(pks[i])), m,
                                    len(msgs[i]), 1 if prehashed else 0, ctx, len(context))
    return out
