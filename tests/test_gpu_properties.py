"""GPU property tests mirroring the reference's own test_ec (test/test_goldilocks.cxx:316-437):
algebraic identities between the scalarmul implementations, invariance under the projective
representation (random Z-scaling, the reference's point_debugging_pscale; 2-torsion "torque",
point_debugging_torque; unreduced limbs), all through the C ABI."""
import numpy as np
import pytest

import _gen
from _libs import P, Q

pytestmark = pytest.mark.gpu
N = 256


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


def limbs_to_int(l):
    return sum(int(x) << (56 * i) for i, x in enumerate(l))


def int_to_limbs(v, sloppy=0):
    """56-bit limbs of v mod p; sloppy=1 adds p (limb-wise, no carry) so limbs exceed 2^56."""
    v %= P
    l = [(v >> (56 * i)) & ((1 << 56) - 1) for i in range(8)]
    if sloppy:
        pl = [(1 << 56) - 1] * 8
        pl[4] -= 1
        l = [a + b for a, b in zip(l, pl)]
    return l


def rescale(points, seed):
    """Same group elements, different representatives: scale (X,Y,Z,T) by random factors, add
    2-torsion to every other point (x,y -> -x,-y), leave limbs unreduced for every third."""
    rng = np.random.default_rng(seed)
    out = np.empty_like(points)
    for i, p in enumerate(points):
        c = [limbs_to_int(p[8 * k:8 * k + 8]) for k in range(4)]
        f = int.from_bytes(rng.bytes(56), "little") % P or 1
        c = [x * f % P for x in c]
        if i % 2:
            c[0], c[1] = (-c[0]) % P, (-c[1]) % P        # torque: (-X, -Y, Z, T)
        out[i] = np.array(sum((int_to_limbs(x, sloppy=(i % 3 == 0)) for x in c), []), dtype=np.uint64)
    return out


def s_int(s):
    return [int.from_bytes(r.tobytes(), "little") for r in s]


@pytest.fixture(scope="module")
def world(ga):
    x, y = _gen.random_scalars(N, b"prop-x"), _gen.random_scalars(N, b"prop-y")
    p = ga.precomputed_scalarmul_batch(_gen.random_scalars(N, b"prop-p"))
    q = ga.precomputed_scalarmul_batch(_gen.random_scalars(N, b"prop-q"))
    return x, y, p, q


def test_representation_invariance(ga, world):
    x, y, p, q = world
    p2 = rescale(p, 1)
    assert (ga.point_encode_batch(p2) == ga.point_encode_batch(p)).all()           # encoding is canonical
    assert (ga.point_encode_batch(ga.point_scalarmul_batch(p2, x)) ==
            ga.point_encode_batch(ga.point_scalarmul_batch(p, x))).all()
    e1 = ga.point_encode_like_eddsa_batch(p)
    # the EdDSA encoding multiplies by 4, which kills the 2-torsion component as well
    assert (ga.point_encode_like_eddsa_batch(p2) == e1).all()
    dec, st = ga.point_decode_batch(ga.point_encode_batch(p))
    assert (st == -1).all() and (ga.point_encode_batch(dec) == ga.point_encode_batch(p)).all()


def test_group_laws_and_scalarmul_identities(ga, world):
    import torch
    x, y, p, q = world
    enc = ga.point_encode_batch
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    def op(a, b, which):
        out = torch.empty((len(a), 32), dtype=torch.int64, device="cuda")
        da, db = d(a), (d(b) if b is not None else None)          # keep the device copies alive across the call
        ga.dev("point_op", out.data_ptr(), da.data_ptr(), db.data_ptr() if db is not None else None, which, len(a), None)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint64)
    pq = op(p, q, 0)
    # x*(p+q) == x*p + x*q
    xp, xq = ga.point_scalarmul_batch(p, x), ga.point_scalarmul_batch(q, x)
    assert (enc(ga.point_scalarmul_batch(pq, x)) == enc(op(xp, xq, 0))).all()
    # (x*y)*p == x*(y*p)
    xy = _gen.scalars_from_ints([a * b for a, b in zip(s_int(x), s_int(y))])
    assert (enc(ga.point_scalarmul_batch(p, xy)) == enc(ga.point_scalarmul_batch(ga.point_scalarmul_batch(p, y), x))).all()
    # x*p + y*q == double_scalarmul
    yq = ga.point_scalarmul_batch(q, y)
    assert (enc(op(xp, yq, 0)) == enc(ga.point_double_scalarmul_batch(p, x, q, y))).all()
    # x*base + y*q == base_double_scalarmul_non_secret
    xb = ga.precomputed_scalarmul_batch(x)
    assert (enc(op(xb, yq, 0)) == enc(ga.point_double_scalarmul_batch(None, x, q, y))).all()
    # times_two, subtraction, p - p = identity
    two = _gen.scalars_from_ints([2] * N)
    assert (enc(op(p, None, 2)) == enc(ga.point_scalarmul_batch(p, two))).all()
    assert (enc(op(pq, q, 1)) == enc(p)).all()
    assert (enc(op(p, p, 1)) == 0).all()
    # Precomputed(p)*x == p*x  (one table, many scalars), and the comb of the base point == window table
    tab = ga.precompute(p[0])
    assert (enc(ga.precomputed_scalarmul_batch(x, table=tab)) == enc(ga.point_scalarmul_batch(np.tile(p[0], (N, 1)), x))).all()
    assert (enc(ga.precomputed_scalarmul_batch(x, table=ga.precomputed_base())) == enc(xb)).all()
    # direct_scalarmul == encode(x * decode(.))
    got, st = ga.direct_scalarmul_batch(enc(p), x)
    assert (st == -1).all() and (got == enc(xp)).all()


def test_scalar_edge_identities(ga, world):
    x, y, p, q = world
    enc = ga.point_encode_batch
    zero, one, minus1 = (_gen.scalars_from_ints([v] * N) for v in (0, 1, Q - 1))
    assert (enc(ga.point_scalarmul_batch(p, zero)) == 0).all()
    assert (enc(ga.point_scalarmul_batch(p, one)) == enc(p)).all()
    neg = ga.point_scalarmul_batch(p, minus1)
    import torch
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    out = torch.empty((N, 32), dtype=torch.int64, device="cuda")
    dp, dn = d(p), d(neg)
    ga.dev("point_op", out.data_ptr(), dp.data_ptr(), dn.data_ptr(), 0, N, None)
    torch.cuda.synchronize()
    assert (enc(out.cpu().numpy().view(np.uint64)) == 0).all()        # p + (-p) = identity
    # q*p = identity: scalars are taken mod q, Q itself reduces to 0 in the callers' encoding
    assert (enc(ga.precomputed_scalarmul_batch(zero)) == 0).all()


def test_negate_cond_sel_destroy(ga, world):
    """point_negate on the GPU ((q-1)*P == -P, P + (-P) == identity) and the memory-only helpers."""
    import ctypes as C
    x, y, p, q = world
    L = ga.lib()
    qm1 = _gen.scalars_from_ints([Q - 1] * N)
    want = ga.point_scalarmul_batch(p, qm1)
    ident = ga.point_identity()
    for i in range(0, N, 37):
        neg = np.empty(32, np.uint64)
        L.goldilocks_448_point_negate(neg.ctypes.data, p[i].ctypes.data)
        assert L.goldilocks_448_point_eq(neg.ctypes.data, want[i].ctypes.data)
        s = np.empty(32, np.uint64)
        L.goldilocks_448_point_add(s.ctypes.data, neg.ctypes.data, p[i].ctypes.data)
        assert L.goldilocks_448_point_eq(s.ctypes.data, ident.ctypes.data)
        L.goldilocks_448_point_negate(neg.ctypes.data, neg.ctypes.data)          # in place
        assert L.goldilocks_448_point_eq(neg.ctypes.data, p[i].ctypes.data)
    out = np.empty(32, np.uint64)
    L.goldilocks_448_point_cond_sel(out.ctypes.data, p[0].ctypes.data, q[0].ctypes.data, 0)
    assert (out == p[0]).all()
    L.goldilocks_448_point_cond_sel(out.ctypes.data, p[0].ctypes.data, q[0].ctypes.data, 1 << 40)
    assert (out == q[0]).all()
    L.goldilocks_448_point_destroy(out.ctypes.data)
    assert not out.any()
    tab = ga.precompute(p[0]).copy()
    L.goldilocks_448_precomputed_destroy(tab.ctypes.data)
    assert not tab.any()
