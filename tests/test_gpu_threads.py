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


def test_per_call_table_access_from_two_threads(ga, O):
    """The table access is a property of the CALL (include/goldilocks_amd.h, the *_ex entry points), the
    process-wide setting only its default: one thread multiplying public scalars with
    GOLDILOCKS_AMD_CALL_TABLES_FAST must not change what a concurrent thread's default-mode signing and
    multiplying run.  goldilocks_amd_thread_mode_counts tallies, per calling thread, what each mode-dependent
    call resolved to when it launched."""
    import torch
    assert ga.get_table_access() == ga.TABLES_INDEX_INDEPENDENT
    n = 20000                                         # above the one-operation-per-wave thresholds: lane kernels
    scal = _gen.stream_scalars(n, b"percall/s")
    bases_h = ga.precomputed_scalarmul_batch(_gen.stream_scalars(n, b"percall/b"))
    d = lambda a, t=np.int64: torch.from_numpy(np.ascontiguousarray(a).view(t)).cuda()
    d_b, d_s = d(bases_h), d(scal)
    sk = np.frombuffer(_gen.stream(b"percall/sk", 57 * n), np.uint8).reshape(n, 57).copy()
    d_sk = d(sk, np.uint8)
    d_pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_derive_public_key", d_pk.data_ptr(), d_sk.data_ptr(), n, None)
    d_msg = d(np.frombuffer(_gen.stream(b"percall/m", 16 * n), np.uint8).reshape(n, 16).copy(), np.uint8)
    torch.cuda.synchronize()
    want_pts = _gen.oracle_encode(_gen.oracle_varbase(O, bases_h[:64], scal[:64]))
    counts, errors = {}, []
    rounds = 12

    def public_side():
        try:
            before = ga.thread_mode_counts()
            out = torch.empty((n, 32), dtype=torch.int64, device="cuda")
            st = torch.cuda.Stream()
            zero = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
            for _ in range(rounds):     # s*b + 0*b through the two-scalar entry point: the one whose FAST is digit-addressed tables
                ga.dev("point_double_scalarmul", out.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), d_b.data_ptr(), zero.data_ptr(), n,
                       st.cuda_stream, flags=ga.CALL_TABLES_FAST)
            st.synchronize()
            after = ga.thread_mode_counts()
            counts["public"] = (after[0] - before[0], after[1] - before[1])
            if not (ga.point_encode_batch(out[:64].cpu().numpy().view(np.uint64)) == want_pts).all():
                errors.append("fast results")
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    def secret_side():
        try:
            before = ga.thread_mode_counts()
            sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
            out = torch.empty((n, 32), dtype=torch.int64, device="cuda")
            st = torch.cuda.Stream()
            for _ in range(rounds):
                ga.dev("ed448_sign", sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 16, 0,
                       None, 0, n, st.cuda_stream)                      # no flags: the library default
                ga.dev("point_scalarmul", out.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), n, st.cuda_stream)
            st.synchronize()
            after = ga.thread_mode_counts()
            counts["secret"] = (after[0] - before[0], after[1] - before[1])
            if not (ga.point_encode_batch(out[:64].cpu().numpy().view(np.uint64)) == want_pts).all():
                errors.append("default-mode results")
            v = torch.empty(n, dtype=torch.int32, device="cuda")
            ga.dev("ed448_verify", v.data_ptr(), sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 16, 0, None, 0, n, None)
            if int((v == -1).sum()) != n:
                errors.append("signatures")
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=public_side), threading.Thread(target=secret_side)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert counts["public"] == (rounds, 0)            # every call of that thread: digit-addressed tables
    assert counts["secret"] == (0, 2 * rounds)        # every call of this one: index-independent, throughout
    assert ga.get_table_access() == ga.TABLES_INDEX_INDEPENDENT   # nobody touched the default
    with pytest.raises(ga.GoldilocksAmdError):
        ga.dev("point_scalarmul", d_b.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), n, None, flags=3)   # unknown flags
