"""Host threads calling the library at once (ctypes releases the GIL for the duration of a call): the per-device
lock serialises the calls that share the workspace and the staging buffers; every result must be what the same
call gives alone (and what the oracle gives)."""
import threading

import numpy as np
import pytest

import _gen

pytestmark = pytest.mark.gpu


def test_concurrent_host_calls_from_four_threads(ga, O):
    n_small, n_big = 300, 140000            # one operation per wave / lane kernels with pipelined copies
    s_small, s_big = _gen.random_scalars(n_small, b"thr/s1"), _gen.random_scalars(n_big, b"thr/s2")
    bases = _gen.oracle_fixed(O, _gen.random_scalars(n_small, b"thr/b"))
    sigs, pks, msgs = _gen.signatures(O, 600, msglen=24, seed=b"thr/sig", nkeys=7)
    sigs[::5, 70] ^= 1
    sk = np.frombuffer(_gen.stream(b"thr/sk", 57 * 200), np.uint8).reshape(200, 57).copy()
    want_var = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s_small))
    want_fixed_small = _gen.oracle_encode(_gen.oracle_fixed(O, s_big[:256]))
    want_st = _gen.oracle_verify(O, sigs, pks, msgs)
    want_pk = ga.ed448_derive_public_key_batch(sk)
    want_sig = ga.ed448_sign_batch(sk, want_pk, [b"m%d" % i for i in range(200)])
    errors = []

    def check(cond, what):
        if not cond:
            errors.append(what)

    def worker(kind):
        try:
            for _ in range(6):
                if kind == 0:
                    check((ga.point_encode_batch(ga.point_scalarmul_batch(bases, s_small)) == want_var).all(), "variable base")
                elif kind == 1:
                    got = ga.precomputed_scalarmul_batch(s_big)
                    check((ga.point_encode_batch(got[:256]) == want_fixed_small).all(), "fixed base, large batch")
                elif kind == 2:
                    check((ga.ed448_verify_batch(sigs, pks, msgs) == want_st).all(), "verify")
                else:
                    pk = ga.ed448_derive_public_key_batch(sk)
                    check((pk == want_pk).all(), "derive")
                    check((ga.ed448_sign_batch(sk, pk, [b"m%d" % i for i in range(200)]) == want_sig).all(), "sign")
        except Exception as e:   # noqa: BLE001
            errors.append("%d: %r" % (kind, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
