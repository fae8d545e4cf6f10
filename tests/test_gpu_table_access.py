"""GPU tests of the table-access policy (reference: constant_time_lookup,
src/include/constant_time.h:134-183, used at src/goldilocks.c:437-442 and :864; contract in the
reference's README.md:92-97).  The library's default is index-independent access for every scalar that
may be secret; GOLDILOCKS_AMD_TABLES_FAST is the opt-in for public scalars.  Both must give the same
bytes -- checked here on the reference's own golden vectors (F1) and against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import _gen

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pt(a):
    from _libs import Point
    return a.ctypes.data_as(C.POINTER(Point))


def _sc(a):
    from _libs import Scalar
    return a.ctypes.data_as(C.POINTER(Scalar))


@pytest.fixture()
def modes(ga):
    """Runs the body once per mode and always leaves the library in its default."""
    def each(fn):
        out = {}
        try:
            for name, mode in (("index_independent", ga.TABLES_INDEX_INDEPENDENT), ("fast", ga.TABLES_FAST)):
                ga.set_table_access(mode)
                assert ga.get_table_access() == mode
                out[name] = fn()
        finally:
            ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
        return out
    return each


def test_default_is_index_independent():
    """A fresh process: the drop-in names keep the reference's constant-time contract unless told otherwise."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import libgoldilocks_amd as ga; "
                        "print(ga.get_table_access() == ga.TABLES_INDEX_INDEPENDENT)" % root],
                       capture_output=True, text=True, timeout=300)
    assert r.stdout.strip() == "True", r.stdout + r.stderr


def test_both_modes_reproduce_golden_f1(ga, modes):
    """The 1024 reference vectors of fixture F1 (variable-base scalarmul) in both table-access modes."""
    d = np.load(os.path.join(G, "f1_varbase.npz"))
    bases, st = ga.point_decode_batch(d["base"], allow_identity=True)
    assert (st == -1).all()
    res = modes(lambda: ga.point_encode_batch(ga.point_scalarmul_batch(bases, d["scalar"])))
    assert (res["index_independent"] == d["out"]).all()
    assert (res["fast"] == d["out"]).all()


def test_both_modes_on_every_variable_base_entry_point(ga, O, modes):
    """direct_scalarmul, point_double_scalarmul, point_dual_scalarmul: both modes against the oracle,
    ragged batch sizes (1, 63, 257) and edge scalars."""
    from _libs import Q
    for n in (1, 63, 257):
        k = _gen.stream_scalars(n, b"ta/base/%d" % n)
        bases = _gen.oracle_fixed(O, k)
        bases2 = _gen.oracle_fixed(O, _gen.stream_scalars(n, b"ta/base2/%d" % n))
        s1 = _gen.stream_scalars(n, b"ta/s1/%d" % n)
        s2 = _gen.stream_scalars(n, b"ta/s2/%d" % n)
        edge = _gen.scalars_from_ints([0, 1, Q - 1, 2**445, 2**444 - 1, 15, 16, 17])
        s1[:min(n, len(edge))] = edge[:min(n, len(edge))]
        want_mul = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s1))
        want_mul2 = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s2))
        dbl = np.empty((n, 32), dtype=np.uint64)
        for i in range(n):
            O.orc_point_double_scalarmul(_pt(dbl[i]), _pt(bases[i]), _sc(s1[i]), _pt(bases2[i]), _sc(s2[i]))
        want_dbl = _gen.oracle_encode(dbl)
        base_enc = _gen.oracle_encode(bases)

        def body():
            enc, st = ga.direct_scalarmul_batch(base_enc, s1)
            o1, o2 = ga.point_dual_scalarmul_batch(bases, s1, s2)
            return dict(direct=enc, direct_st=st, dbl=ga.point_encode_batch(ga.point_double_scalarmul_batch(bases, s1, bases2, s2)),
                        dual1=ga.point_encode_batch(o1), dual2=ga.point_encode_batch(o2),
                        mul=ga.point_encode_batch(ga.point_scalarmul_batch(bases, s1)))
        for mode, r in modes(body).items():
            assert (r["direct_st"] == -1).all(), mode
            assert (r["direct"] == want_mul).all(), (mode, "direct", n)
            assert (r["mul"] == want_mul).all(), (mode, "scalarmul", n)
            assert (r["dbl"] == want_dbl).all(), (mode, "double_scalarmul", n)
            assert (r["dual1"] == want_mul).all() and (r["dual2"] == want_mul2).all(), (mode, "dual", n)


def test_exceptional_bases_and_scalars_in_both_modes(ga, O, modes):
    """What the table-free ladder (csrc/montgomery.hpp) treats by select, through every variable-base entry point
    that has a ladder, in both modes and through both dispatch paths (one operation per wave / per lane): the
    identity and the 2-torsion point (0, -1) as bases; bases with a 2-torsion component P + (0, -1) = (-x, -y)
    (order 2q: the same class as s P); scalars 0, 1, q - 1 (where (s + 1) P is the identity and the recovery of
    the second coordinate degenerates), q - 2, and scalars that are not reduced (q + 5, 2 q + 1 as raw words)."""
    from _libs import Q, P, Gf
    vals = [0, 1, 2, Q - 1, Q - 2, (Q + 1) // 2, 2**445, 5]
    n = 4 * len(vals)
    scal = np.concatenate([_gen.scalars_from_ints(vals)] * 4)
    pts = _gen.oracle_fixed(O, _gen.stream_scalars(len(vals), b"ta/exc/base"))
    ident = np.zeros(32, np.uint64); ident[8] = 1; ident[16] = 1
    t2 = ident.copy(); t2[8:16] = np.frombuffer(Gf.from_int(P - 1), np.uint64)
    shifted = pts.copy()
    for i in range(len(vals)):
        for fld in (0, 8):
            v = sum(int(shifted[i][fld + k]) << (56 * k) for k in range(8)) % P
            shifted[i][fld:fld + 8] = np.frombuffer(Gf.from_int((P - v) % P), np.uint64)
    bases = np.concatenate([pts, shifted, np.repeat(ident.reshape(1, 32), len(vals), 0), np.repeat(t2.reshape(1, 32), len(vals), 0)])
    raw = np.empty((2, 7), np.uint64)
    raw[0] = np.frombuffer((Q + 5).to_bytes(56, "little"), np.uint64)
    raw[1] = np.frombuffer((2 * Q + 1).to_bytes(56, "little"), np.uint64)
    bases = np.concatenate([bases, pts[:2]]); scal = np.concatenate([scal, raw]); n += 2
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases, scal))
    assert (want[2 * len(vals):4 * len(vals)] == 0).all()                   # the identity's class
    assert (want[:len(vals)] == want[len(vals):2 * len(vals)]).all()        # the 2-torsion shift changes nothing
    s2 = _gen.stream_scalars(n, b"ta/exc/s2")
    saved = ga.get_wave_batch_max()
    try:
        for wave_max in (saved, 0):
            ga.set_wave_batch_max(wave_max)

            def body():
                o1, o2 = ga.point_dual_scalarmul_batch(bases, scal, s2)
                dbl = ga.point_double_scalarmul_batch(bases, scal, bases, s2)      # s P + s2 P = (s + s2) P
                return dict(mul=ga.point_encode_batch(ga.point_scalarmul_batch(bases, scal)), dual1=ga.point_encode_batch(o1),
                            dual2=ga.point_encode_batch(o2), dbl=ga.point_encode_batch(dbl))
            sums = np.empty((n, 7), np.uint64)
            for i in range(n):
                a = int.from_bytes(scal[i].tobytes(), "little"); b = int.from_bytes(s2[i].tobytes(), "little")
                sums[i] = np.frombuffer(((a + b) % Q).to_bytes(56, "little"), np.uint64)
            want_sum = _gen.oracle_encode(_gen.oracle_varbase(O, bases, sums))
            want2 = _gen.oracle_encode(_gen.oracle_varbase(O, bases, s2))
            for mode, r in modes(body).items():
                assert (r["mul"] == want).all(), (mode, wave_max, np.nonzero((r["mul"] != want).any(1))[0])
                assert (r["dual1"] == want).all() and (r["dual2"] == want2).all(), (mode, wave_max)
                assert (r["dbl"] == want_sum).all(), (mode, wave_max)
    finally:
        ga.set_wave_batch_max(saved)


def test_index_independent_scan_on_a_full_residency(ga, O):
    """More operations than resident lanes (grid-stride rounds, partial last wave) through the table-free
    ladder; a sample of lanes against the oracle and all of them against the fast tables on the device."""
    import torch
    info = ga.device_info()
    n = info["compute_units"] * 2 * 256 + 1000 + 37
    k = torch.from_numpy(_gen.stream_scalars(n, b"ta/full/base").view(np.int64)).cuda()
    s = torch.from_numpy(_gen.stream_scalars(n, b"ta/full/scalar").view(np.int64)).cuda()
    bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    out_ct, out_fast = torch.empty_like(bases), torch.empty_like(bases)
    ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
    try:
        ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
        ga.dev("point_scalarmul", out_ct.data_ptr(), bases.data_ptr(), s.data_ptr(), n, None)
        # (one scalar times a variable base runs the ladder in both modes since round 6: the every-lane cross-check is the
        # two-scalar kernel with its digit-addressed tables, s*P + 0*P)
        ga.set_table_access(ga.TABLES_FAST)
        zero = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
        ga.dev("point_double_scalarmul", out_fast.data_ptr(), bases.data_ptr(), s.data_ptr(), bases.data_ptr(), zero.data_ptr(), n, None)
    finally:
        ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", st.data_ptr(), out_ct.data_ptr(), out_fast.data_ptr(), 0, n, None)
    assert int((st == -1).sum()) == n
    idx = np.unique(np.concatenate([np.arange(0, n, 4099), [0, 63, 64, n - 1, n - 37, n - 38]]))
    b_h = bases.cpu().numpy().view(np.uint64)[idx]
    s_h = s.cpu().numpy().view(np.uint64)[idx]
    want = _gen.oracle_encode(_gen.oracle_varbase(O, b_h, s_h))
    got = ga.point_encode_batch(out_ct.cpu().numpy().view(np.uint64)[idx])
    assert (got == want).all()


def test_ladder_kernels_over_several_launches_of_one_call(ga, O):
    """A launch of the table-free ladder kernels covers at most 8 operations per resident lane (their shared
    inversions park per-operation state in a bounded workspace), so a call with more than 2^20 operations is cut into
    sub-batches: 2^20 + 70 001 variable-base multiplications in place (out aliases base), dual and double
    multiplications of the same size class, every lane against the digit-addressed kernels on the device and a sample
    against the oracle."""
    import torch
    n = (1 << 20) + 70001
    k = torch.from_numpy(_gen.stream_scalars(n, b"ta/multi/base").view(np.int64)).cuda()
    s = torch.from_numpy(_gen.stream_scalars(n, b"ta/multi/scalar").view(np.int64)).cuda()
    bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
    inplace = bases.clone()
    out_fast = torch.empty_like(bases)
    ga.dev("point_scalarmul", inplace.data_ptr(), inplace.data_ptr(), s.data_ptr(), n, None, flags=ga.CALL_TABLES_INDEX_INDEPENDENT)
    zero = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
    ga.dev("point_double_scalarmul", out_fast.data_ptr(), bases.data_ptr(), s.data_ptr(), bases.data_ptr(), zero.data_ptr(), n, None,
           flags=ga.CALL_TABLES_FAST)           # s*P + 0*P through the digit-addressed two-scalar kernel
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_pred", st.data_ptr(), inplace.data_ptr(), out_fast.data_ptr(), 0, n, None)
    assert int((st == -1).sum()) == n
    idx = np.unique(np.concatenate([np.arange(0, n, 65537), [0, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, n - 1]]))
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases.cpu().numpy().view(np.uint64)[idx], s.cpu().numpy().view(np.uint64)[idx]))
    assert (ga.point_encode_batch(inplace.cpu().numpy().view(np.uint64)[idx]) == want).all()
    # s P + k P through the two-ladder kernel == (s + k) P through the digit-addressed one
    dbl = torch.empty_like(bases)
    ga.dev("point_double_scalarmul", dbl.data_ptr(), bases.data_ptr(), s.data_ptr(), bases.data_ptr(), k.data_ptr(), n, None,
           flags=ga.CALL_TABLES_INDEX_INDEPENDENT)
    o1, o2 = torch.empty_like(bases), torch.empty_like(bases)
    ga.dev("point_dual_scalarmul", o1.data_ptr(), o2.data_ptr(), bases.data_ptr(), s.data_ptr(), k.data_ptr(), n, None,
           flags=ga.CALL_TABLES_INDEX_INDEPENDENT)
    ga.dev("point_pred", st.data_ptr(), o1.data_ptr(), out_fast.data_ptr(), 0, n, None)
    assert int((st == -1).sum()) == n
    summed = torch.empty_like(bases)
    ga.dev("point_op", summed.data_ptr(), o1.data_ptr(), o2.data_ptr(), 0, n, None)          # s P + k P by point_add
    ga.dev("point_pred", st.data_ptr(), dbl.data_ptr(), summed.data_ptr(), 0, n, None)
    assert int((st == -1).sum()) == n


@pytest.mark.parametrize("mode", ["index-independent", "fast"])
def test_two_scalar_device_entries_with_every_output_aliasing(ga, O, mode):
    """goldilocks_amd_point_double_scalarmul_dev / _point_dual_scalarmul_dev with each output aliasing each input
    array (the reference computes into temporaries and allows all of it, src/goldilocks.c:467-541, :543-642) above
    the wave-kernel threshold, several operations per lane: the results must equal the un-aliased call's, and a
    sample the oracle's."""
    import torch
    flags = ga.CALL_TABLES_INDEX_INDEPENDENT if mode == "index-independent" else ga.CALL_TABLES_FAST
    n = ga.device_info()["compute_units"] * 2 * 256 * 2 + 333
    k1 = torch.from_numpy(_gen.stream_scalars(n, b"alias2/b1").view(np.int64)).cuda()
    k2 = torch.from_numpy(_gen.stream_scalars(n, b"alias2/b2").view(np.int64)).cuda()
    s1 = torch.from_numpy(_gen.stream_scalars(n, b"alias2/s1").view(np.int64)).cuda()
    s2 = torch.from_numpy(_gen.stream_scalars(n, b"alias2/s2").view(np.int64)).cuda()
    b1 = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    b2 = torch.empty_like(b1)
    ga.dev("precomputed_scalarmul", b1.data_ptr(), None, k1.data_ptr(), n, None)
    ga.dev("precomputed_scalarmul", b2.data_ptr(), None, k2.data_ptr(), n, None)
    st = torch.empty(n, dtype=torch.int32, device="cuda")

    def same(x, y):
        ga.dev("point_pred", st.data_ptr(), x.data_ptr(), y.data_ptr(), 0, n, None)
        return int((st == -1).sum()) == n

    # double: combo = s1 b1 + s2 b2
    ref = torch.empty_like(b1)
    ga.dev("point_double_scalarmul", ref.data_ptr(), b1.data_ptr(), s1.data_ptr(), b2.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    idx = np.unique(np.concatenate([np.arange(0, n, 9973), [0, 63, 64, n - 1]]))
    h = lambda t: t.cpu().numpy().view(np.uint64)[idx]
    want = _gen.oracle_encode(_gen.oracle_double(O, h(b1), h(s1), h(b2), h(s2)))
    assert (ga.point_encode_batch(h(ref)) == want).all()
    x = b1.clone()                                           # out == b1
    ga.dev("point_double_scalarmul", x.data_ptr(), x.data_ptr(), s1.data_ptr(), b2.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert same(x, ref)
    x = b2.clone()                                           # out == b2
    ga.dev("point_double_scalarmul", x.data_ptr(), b1.data_ptr(), s1.data_ptr(), x.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert same(x, ref)
    ref11 = torch.empty_like(b1)                             # out == b1 == b2
    ga.dev("point_double_scalarmul", ref11.data_ptr(), b1.data_ptr(), s1.data_ptr(), b1.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    x = b1.clone()
    ga.dev("point_double_scalarmul", x.data_ptr(), x.data_ptr(), s1.data_ptr(), x.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert same(x, ref11)
    # dual: (a1, a2) = (s1 b, s2 b)
    r1, r2 = torch.empty_like(b1), torch.empty_like(b1)
    ga.dev("point_dual_scalarmul", r1.data_ptr(), r2.data_ptr(), b1.data_ptr(), s1.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert (ga.point_encode_batch(h(r1)) == _gen.oracle_encode(_gen.oracle_varbase(O, h(b1), h(s1)))).all()
    assert (ga.point_encode_batch(h(r2)) == _gen.oracle_encode(_gen.oracle_varbase(O, h(b1), h(s2)))).all()
    x, y = b1.clone(), torch.empty_like(b1)                  # a1 == base
    ga.dev("point_dual_scalarmul", x.data_ptr(), y.data_ptr(), x.data_ptr(), s1.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert same(x, r1) and same(y, r2)
    x, y = torch.empty_like(b1), b1.clone()                  # a2 == base
    ga.dev("point_dual_scalarmul", x.data_ptr(), y.data_ptr(), y.data_ptr(), s1.data_ptr(), s2.data_ptr(), n, None, flags=flags)
    assert same(x, r1) and same(y, r2)


def test_one_scalar_times_a_variable_base_holds_no_tables_in_either_mode(ga, O):
    """Round 6 retired the digit-addressed per-lane tables of goldilocks_448_point_scalarmul / _direct_scalarmul (544 MiB
    of workspace, 46 x the algorithmic traffic, no faster than the ladder): a call with GOLDILOCKS_AMD_CALL_TABLES_FAST
    runs the ladder too -- the workspace after 2^20 multiplications stays at the ladder's 128 MiB and the results are
    the oracle's."""
    import torch
    ga.release_memory()
    n = 1 << 20
    k = torch.from_numpy(_gen.stream_scalars(n, b"ta/retired/base").view(np.int64)).cuda()
    s = torch.from_numpy(_gen.stream_scalars(n, b"ta/retired/scalar").view(np.int64)).cuda()
    bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
    ga.release_memory()
    out = torch.empty_like(bases)
    ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), s.data_ptr(), n, None, flags=ga.CALL_TABLES_FAST)
    torch.cuda.synchronize()
    assert 0 < ga.device_info()["workspace_bytes"] <= 128 << 20
    enc_in = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
    enc_out, st = torch.empty_like(enc_in), torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("point_encode", enc_in.data_ptr(), bases.data_ptr(), n, None)
    ga.dev("direct_scalarmul", enc_out.data_ptr(), st.data_ptr(), enc_in.data_ptr(), s.data_ptr(), 0, 0, n, None, flags=ga.CALL_TABLES_FAST)
    torch.cuda.synchronize()
    assert ga.device_info()["workspace_bytes"] <= 128 << 20 and int((st == -1).sum()) == n
    idx = np.unique(np.concatenate([np.arange(0, n, 9973), [0, 63, 64, n - 1]]))
    want = _gen.oracle_encode(_gen.oracle_varbase(O, bases.cpu().numpy().view(np.uint64)[idx], s.cpu().numpy().view(np.uint64)[idx]))
    assert (ga.point_encode_batch(out.cpu().numpy().view(np.uint64)[idx]) == want).all()
    assert (enc_out.cpu().numpy()[idx] == want).all()
