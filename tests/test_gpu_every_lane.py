"""EVERY lane of the kernels that live at the register limit, against the oracle.

Round 5 found a block that the compiler got wrong inside a kernel with 170 - 210 spilled registers (docs/history/r05.md H;
tools/probes/miscompile_r05_repro.py asks a toolchain whether it still does).  The kernels of that shape -- more than 100
spilled vector registers in tools/kernel_resources.py: k_point_dual_scalarmul_ct, k_double_scalarmul(_ct),
k_direct_scalarmul_ct, k_build_bwt -- are run here at one full residency of the device plus a ragged tail, with the inputs
their divergent blocks exist for interleaved among ordinary lanes (the identity and the 2-torsion point as bases, scalars
0, 1, q - 1, 2^445, encodings that do not decode or decode to the identity), and EVERY lane is compared with the reference's
function of that name as the oracle restates it (one call per operation, src/goldilocks.c:467-541, 543-642, 888-903); the
window table of the base point is compared entry by entry.  Bit-exact on encodings / status words / canonical bytes."""
import ctypes as C
import os

import numpy as np
import pytest

import _gen
from _libs import P, Q

pytestmark = pytest.mark.gpu
NT = _gen.NTHREADS


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _limbs(v):
    return np.frombuffer((v % P).to_bytes(56, "little") + b"\0" * 8, np.uint8)[:56]


def _point(x, y, z=1, t=None):
    """a point_s from affine-looking integers: 4 x (8 limbs of 56 bits)"""
    t = x * y if t is None else t
    out = np.zeros(32, np.uint64)
    for c, v in enumerate((x, y, z, t)):
        v %= P
        for i in range(8):
            out[8 * c + i] = (v >> (56 * i)) & ((1 << 56) - 1)
    return out


def _inputs(ga, O, n, label):
    """n (point, scalar, scalar) triples: random points k*B, with the identity, the 2-torsion point (0, -1) and the base
    point itself every 97 lanes, and the scalars 0, 1, q - 1, 2^445 every 89 lanes (in both scalar arrays, out of step)"""
    k = _gen.stream_scalars(n, label + b"/k")
    pts = _gen.oracle_fixed(O, k)
    special_pts = [_point(0, 1), _point(0, P - 1), ga.point_base().reshape(32)]
    for j, i in enumerate(range(5, n, 97)):
        pts[i] = special_pts[j % len(special_pts)]
    s1 = _gen.stream_scalars(n, label + b"/s1")
    s2 = _gen.stream_scalars(n, label + b"/s2")
    special_sc = _gen.scalars_from_ints([0, 1, Q - 1, 1 << 445])
    for j, i in enumerate(range(11, n, 89)):
        s1[i] = special_sc[j % 4]
        s2[(i + 40) % n] = special_sc[(j + 1) % 4]
    return pts, s1, s2


def _size(ga):
    return ga.device_info()["compute_units"] * 2 * 256 + 293        # one full residency of the lane kernels + a ragged tail


@pytest.mark.parametrize("mode", ["index_independent", "fast"])
def test_every_lane_of_double_scalarmul(ga, O, mode):
    """k_double_scalarmul_ct (two ladders, 131 spilled registers) / k_double_scalarmul (two window tables on one doubling
    chain, 146): s1*b1 + s2*b2, src/goldilocks.c:467-541"""
    import torch
    n = _size(ga)
    b1, s1, s2 = _inputs(ga, O, n, b"el/double/1")
    b2, _, _ = _inputs(ga, O, n, b"el/double/2")
    b2 = np.roll(b2, 31, axis=0)                                    # the special points meet ordinary partners, and each other
    want = np.empty((n, 32), np.uint64)
    O.orc_point_double_scalarmul_batch(_p(want), _p(b1), _p(s1), _p(b2), _p(s2), n, NT)
    d = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    out = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    flags = ga.CALL_TABLES_FAST if mode == "fast" else ga.CALL_TABLES_INDEX_INDEPENDENT
    db1, ds1, db2, ds2 = d(b1), d(s1), d(b2), d(s2)
    ga.dev("point_double_scalarmul", out.data_ptr(), db1.data_ptr(), ds1.data_ptr(), db2.data_ptr(), ds2.data_ptr(), n, None, flags=flags)
    got = out.cpu().numpy().view(np.uint64)
    bad = np.nonzero((ga.point_encode_batch(got) != _gen.oracle_encode(want)).any(axis=1))[0]
    assert len(bad) == 0, (mode, bad[:20])


@pytest.mark.parametrize("mode", ["index_independent", "fast"])
def test_every_lane_of_dual_scalarmul(ga, O, mode):
    """k_point_dual_scalarmul_ct (174 spilled registers; both table-access modes run it since round 6): (s1*b, s2*b),
    src/goldilocks.c:543-642"""
    import torch
    n = _size(ga)
    b, s1, s2 = _inputs(ga, O, n, b"el/dual")
    w1, w2 = np.empty((n, 32), np.uint64), np.empty((n, 32), np.uint64)
    O.orc_point_dual_scalarmul_batch(_p(w1), _p(w2), _p(b), _p(s1), _p(s2), n, NT)
    d = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    o1 = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    o2 = torch.empty_like(o1)
    flags = ga.CALL_TABLES_FAST if mode == "fast" else ga.CALL_TABLES_INDEX_INDEPENDENT
    db, ds1, ds2 = d(b), d(s1), d(s2)
    ga.dev("point_dual_scalarmul", o1.data_ptr(), o2.data_ptr(), db.data_ptr(), ds1.data_ptr(), ds2.data_ptr(), n, None, flags=flags)
    for got, want in ((o1, w1), (o2, w2)):
        bad = np.nonzero((ga.point_encode_batch(got.cpu().numpy().view(np.uint64)) != _gen.oracle_encode(want)).any(axis=1))[0]
        assert len(bad) == 0, (mode, bad[:20])


@pytest.mark.parametrize("allow_identity,short_circuit", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_every_lane_of_direct_scalarmul(ga, O, allow_identity, short_circuit):
    """k_direct_scalarmul_ct (171 spilled registers; the kernel of round 5's miscompiled block): decode, ladder, encode --
    encodings that do not decode (all ones, an odd s, a value >= p) and the identity's, every 53 lanes, under both identity
    rules and both short-circuit rules (src/goldilocks.c:888-903: without the short circuit the base point is multiplied
    instead and the call still reports failure)"""
    import torch
    n = _size(ga)
    pts, s, _ = _inputs(ga, O, n, b"el/direct")
    enc = _gen.oracle_encode(pts)
    bad_encodings = [np.full(56, 0xff, np.uint8), None, _limbs(P), np.zeros(56, np.uint8), _limbs(P + 2)]
    for j, i in enumerate(range(7, n, 53)):
        e = bad_encodings[j % len(bad_encodings)]
        if e is None:
            enc[i, 0] |= 1                                          # an odd s: "negative", never decodes
        else:
            enc[i] = e
    want, wst = np.zeros((n, 56), np.uint8), np.empty(n, np.int32)
    O.orc_direct_scalarmul_batch(_p(want), _p(wst), _p(enc), _p(s), allow_identity, short_circuit, n, NT)
    denc, ds = torch.from_numpy(enc).cuda(), torch.from_numpy(s.view(np.int64)).cuda()
    out = torch.zeros((n, 56), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("direct_scalarmul", out.data_ptr(), st.data_ptr(), denc.data_ptr(), ds.data_ptr(), allow_identity, short_circuit, n, None)
    got, gst = out.cpu().numpy(), st.cpu().numpy()
    assert (gst == wst).all(), np.nonzero(gst != wst)[0][:20]
    assert (wst == 0).sum() >= n // 53 // 3                         # the failing lanes are there (the identity's encoding decodes when allowed)
    live = (wst == -1) | (short_circuit == 0)                       # a short-circuited lane's output is not defined (it stays untouched)
    bad = np.nonzero((got != want).any(axis=1) & live)[0]
    assert len(bad) == 0, bad[:20]


@pytest.mark.parametrize("bits", [16, 20])
def test_every_entry_of_the_base_points_window_table(ga, O, bits):
    """k_build_bwt (109 spilled registers): the table of `bits`-bit digits, entry e = i 2^(bits-1) + k = ((2k+1) 2^(bits i) mod
    q) * B as an affine niels -- ALL 917 504 entries at 16 bits; at the library's default 20 bits (12 M entries, 2.2 GiB) the
    first and last 4 096 entries of every window and 64 random runs of 1 024 -- against the oracle's
    orc_base_table_entries (the reference's scalar_mul, precomputed_scalarmul and gf_invert), as canonical bytes."""
    try:
        ga.set_base_table_bits(bits)
        per_window, windows = 1 << (bits - 1), (446 + bits - 1) // bits
        total = per_window * windows
        if bits == 16:
            runs = [(0, total)]
        else:
            rng = np.random.default_rng(20)
            runs = [(w * per_window, 4096) for w in range(windows)] + [((w + 1) * per_window - 4096, 4096) for w in range(windows)]
            runs += [(int(f), 1024) for f in rng.integers(0, total - 1024, 64)]
        for first, count in runs:
            got, want = np.empty((count, 168), np.uint8), np.empty((count, 168), np.uint8)
            assert ga.lib().goldilocks_amd_base_table_export(_p(got), first, count) == 0
            O.orc_base_table_entries(_p(want), bits, first, count, NT)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert len(bad) == 0, (bits, first, bad[:20])
        assert ga.get_base_table_bits() == bits
    finally:
        ga.set_base_table_bits(0)
        ga.release_memory()
