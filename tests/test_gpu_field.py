"""GPU parity, field level: every opcode of goldilocks_amd_field_op_dev against
the reference-captured fixture F4 and against the oracle / exact integer arithmetic, EVERY lane
compared.  Covers what only the host-side checker build exercised before: gf_mulw, gf_add / gf_sub with
bias, weak_reduce, eq, lobit, serialize, deserialize with the >= p reject, and fe_mul / fe_sqr on
operands at the limits of the device arithmetic's magnitude contract (gf28.hpp), through the DEVICE
compile of the same code (ref: src/arch_ref64/f_impl.c:168-190, f_impl.h:10-38, src/f_generic.c:19-131)."""
import ctypes as C
import os

import numpy as np
import pytest

import _gen
from _libs import Gf, P

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK56 = (1 << 56) - 1
OP = dict(mul=0, sqr=1, isr=2, strong=3, mulw=4, add=5, sub=6, weak=7, eq=8, lobit=9, ser=10, deser=11, mulmag=12, sqrmag=13, smul=14, ssqr=15)


def val(limbs):
    return sum(int(l) << (56 * i) for i, l in enumerate(limbs)) % P


def raw(limbs):
    return sum(int(l) << (56 * i) for i, l in enumerate(limbs))


def limbs_of(v):
    return [(v >> (56 * i)) & MASK56 for i in range(8)]


def run(ga, op, a, b=None, want_out=True, want_status=False):
    import torch
    n = len(a)
    da = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()
    db = None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint64).view(np.int64)).cuda()
    out = torch.zeros((n, 8), dtype=torch.int64, device="cuda") if want_out else None
    st = torch.zeros(n, dtype=torch.int32, device="cuda") if want_status else None
    ga.dev("field_op", out.data_ptr() if want_out else None, st.data_ptr() if want_status else None, da.data_ptr(),
           db.data_ptr() if db is not None else None, op, n, None)
    torch.cuda.synchronize()
    return (out.cpu().numpy().view(np.uint64) if want_out else None), (st.cpu().numpy() if want_status else None)


def sample(n, seed):
    """Field elements in the ABI limb form: random reduced, weakly reduced with excess bits, and specials."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    a[0] = 0
    a[1] = MASK56                                                   # 2^448 - 1 = 2^224 mod p
    a[2] = [MASK56] * 4 + [MASK56 - 1] + [MASK56] * 3               # p itself
    a[3] = [MASK56 - 1] + [MASK56] * 3 + [MASK56 - 1] + [MASK56] * 3  # p - 1
    a[4] = [1, 0, 0, 0, 0, 0, 0, 0]
    a[5] = [2**56 + 255] * 8                                        # weakly reduced, excess in every limb
    a[6] = [2**57 + 12345] * 8                                      # sum of two elements, unreduced
    a[7] = [0] * 4 + [1] + [0] * 3                                  # phi = 2^224
    a[8:n // 4] += rng.integers(0, 2**20, size=(n // 4 - 8, 8), dtype=np.uint64)   # small excess above 2^56
    return a


def test_fixture_f4_on_the_device(ga):
    """F4 (512 reference-captured cases incl. unreduced limbs): mul, sqr, isr + mask, serialize, all lanes."""
    d = np.load(os.path.join(G, "f4_field.npz"))
    a, b = d["a_limbs"], d["b_limbs"]
    def ser(limbs):
        out, _ = run(ga, OP["ser"], limbs)
        return out[:, :7].copy().view(np.uint8).reshape(len(limbs), 56)
    assert (ser(a) == d["a"]).all() and (ser(b) == d["b"]).all()
    out, _ = run(ga, OP["mul"], a, b)
    assert (ser(out) == d["mul"]).all()
    out, _ = run(ga, OP["sqr"], a)
    assert (ser(out) == d["sqr"]).all()
    out, st = run(ga, OP["isr"], a, want_status=True)
    assert (ser(out) == d["isr"]).all()
    assert ((st != 0).astype(np.uint8) == d["isr_mask"]).all()


def test_mulw_add_sub_weak_every_lane(ga, O):
    n = 2048
    a, b = sample(n, 11), sample(n, 12)[::-1].copy()
    rng = np.random.default_rng(13)
    w = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    w[:6] = [0, 1, 2 * 39082, 39081, 156328, 2**32 - 1]            # the constants the curve code uses
    bw = np.zeros((n, 8), dtype=np.uint64)
    bw[:, 0] = w | (rng.integers(0, 2**31, size=n, dtype=np.uint64) << np.uint64(32))   # high half must be ignored
    out, _ = run(ga, OP["mulw"], a, bw)
    for i in range(n):
        assert val(out[i]) == val(a[i]) * int(w[i]) % P, ("mulw", i)
        go, gi = Gf(), Gf(); gi.limb[:] = [int(x) for x in a[i]]
        O.orc_gf_mulw(C.byref(go), C.byref(gi), int(w[i]))
        assert val(out[i]) == go.value(), ("mulw vs oracle", i)
    for name, f in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y)):
        out, _ = run(ga, OP[name], a, b)
        for i in range(n):
            assert val(out[i]) == f(val(a[i]), val(b[i])) % P, (name, i)
            assert all(int(l) < 2**56 + 2**33 for l in out[i]), (name, "weakly reduced", i)
            go, gx, gy = Gf(), Gf(), Gf()
            gx.limb[:] = [int(x) for x in a[i]]; gy.limb[:] = [int(x) for x in b[i]]
            getattr(O, "orc_gf_" + name)(C.byref(go), C.byref(gx), C.byref(gy))
            assert val(out[i]) == go.value(), (name + " vs oracle", i)
    big = a.copy()
    big[:, :] += np.uint64(5) << np.uint64(56)                      # limbs up to 6 * 2^56: sums of six elements
    out, _ = run(ga, OP["weak"], big)
    for i in range(n):
        assert val(out[i]) == val(big[i]), ("weak", i)
        assert all(int(l) < 2**56 + 2**33 for l in out[i]), ("weak bound", i)


def test_eq_lobit_serialize_deserialize_every_lane(ga, O):
    n = 1024
    a = sample(n, 21)
    b = a.copy()
    rng = np.random.default_rng(22)
    # equal values in different representations (add p limb-wise), and near misses
    p_l = np.array(limbs_of(P), dtype=np.uint64)
    b[::2] += p_l
    b[1::4, 0] ^= np.uint64(1)
    _, st = run(ga, OP["eq"], a, b, want_out=False, want_status=True)
    for i in range(n):
        assert (st[i] != 0) == (val(a[i]) == val(b[i])), ("eq", i)
    _, st = run(ga, OP["lobit"], a, want_out=False, want_status=True)
    for i in range(n):
        assert (st[i] != 0) == bool(val(a[i]) & 1), ("lobit", i)
    out, _ = run(ga, OP["ser"], a)
    by = out[:, :7].copy().view(np.uint8).reshape(n, 56)
    ser = (C.c_uint8 * 56)()
    for i in range(n):
        assert int.from_bytes(by[i].tobytes(), "little") == val(a[i]), ("serialize", i)
        g = Gf(); g.limb[:] = [int(x) for x in a[i]]
        O.orc_gf_serialize(ser, C.byref(g))
        assert bytes(ser) == by[i].tobytes(), ("serialize vs oracle", i)
    # deserialize: canonical strings, and the rejects: p, p + 1, 2^448 - 1, random values >= p
    vals = [0, 1, P - 1, P, P + 1, 2**448 - 1, 2**447, 2**224, 2**224 - 1] + \
           [int(rng.integers(0, 2**62)) | (int(rng.integers(0, 2**62)) << 386) for _ in range(200)] + \
           [P + int(rng.integers(0, 2**60)) for _ in range(50)]
    raw_in = np.zeros((len(vals), 8), dtype=np.uint64)
    for i, v in enumerate(vals):
        raw_in[i, :7] = np.frombuffer(v.to_bytes(56, "little"), dtype=np.uint64)
    out, st = run(ga, OP["deser"], raw_in, want_status=True)
    for i, v in enumerate(vals):
        assert (st[i] != 0) == (v < P), ("deserialize mask", hex(v))
        assert raw(out[i]) == v, ("deserialize limbs", hex(v))
        g = Gf()
        m = O.orc_gf_deserialize(C.byref(g), (C.c_uint8 * 56).from_buffer_copy(v.to_bytes(56, "little")), 0)
        assert (m != 0) == (st[i] != 0)


def test_products_at_the_magnitude_limits(ga):
    """The magnitude contract of gf28.hpp (38 * |a| * |b| + 2^37 < 2^64 per finished column) at its
    documented limits, through the device build: all-ones limbs are the worst case for every column."""
    n = 512
    a = sample(n, 31)
    a[9] = MASK56 + 15                                              # above the weakly-reduced maximum
    b = sample(n, 32)
    b[9] = MASK56 + 15
    b[10] = MASK56
    a[10] = MASK56
    # mag(a) * mag(b) <= 6.7, mag(a) <= 7, mag(b) <= 5: the same pairs the host checker build traps on
    for ma, mb in ((1, 1), (2, 2), (5, 1), (1, 5), (4, 1), (3, 1), (1, 4), (2, 3), (3, 2), (6, 1)):
        out, _ = run(ga, OP["mulmag"] | ma << 8 | mb << 16, a, b)
        for i in range(n):
            assert val(out[i]) == val(a[i]) * val(b[i]) * ma * mb % P, ("mul", ma, mb, i)
    for ma in (1, 2):
        out, _ = run(ga, OP["sqrmag"] | ma << 8, a)
        for i in range(n):
            assert val(out[i]) == (val(a[i]) * ma) ** 2 % P, ("sqr", ma, i)
    with pytest.raises(ga.GoldilocksAmdError):
        run(ga, OP["mulmag"] | 9 << 8 | 1 << 16, a, b)
    with pytest.raises(ga.GoldilocksAmdError):
        run(ga, 16, a, b)


def test_signed_paired_layer_at_its_limits_on_the_device(ga):
    """csrc/gf28s.hpp -- the field layer of the ladders -- through the DEVICE build, where its pair-wise additions are
    inline v_lshl_add_u64 (the host checker runs plain C for them): products and squares of pairable and of negated
    operands at the documented limits (mag(a) * mag(b) <= 3, the square of a sum of two products), all-ones limbs
    among them, every lane against exact integers."""
    n = 512
    a = sample(n, 41)
    b = sample(n, 42)
    a[9] = MASK56; b[9] = MASK56; a[10] = MASK56; b[11] = MASK56
    sgn = lambda k: -(k & 0x7f) if k & 0x80 else k
    for ka, kb in ((1, 1), (2, 1), (1, 2), (3, 1), (2, 0x81), (0x81, 2), (0x81, 0x81), (0x82, 1), (1, 0x82), (3, 0x81), (0x83, 1)):
        out, _ = run(ga, OP["smul"] | ka << 8 | kb << 16, a, b)
        for i in range(n):
            assert val(out[i]) == val(a[i]) * val(b[i]) * sgn(ka) * sgn(kb) % P, ("smul", ka, kb, i)
    for ka, sum2 in ((1, 0), (0x81, 0), (1, 1), (2, 1)):
        out, _ = run(ga, OP["ssqr"] | ka << 8 | sum2 << 16, a)
        for i in range(n):
            assert val(out[i]) == (val(a[i]) * sgn(ka)) ** 2 % P, ("ssqr", ka, sum2, i)
    # beyond the contract (mag(b) <= 2, mag(a) * mag(b) <= 3) the hook refuses instead of returning a wrapped product
    for ka, kb in ((4, 1), (1, 3), (0x81, 0x83), (2, 2), (3, 2), (3, 3)):
        with pytest.raises(ga.GoldilocksAmdError):
            run(ga, OP["smul"] | ka << 8 | kb << 16, a, b)


def test_half_size_pair_of_verification(ga):
    """csrc/lattice.hpp on the device (its quotient estimate is a v_rcp_f64 there, an exact division in the
    host checker): exactly the first pair below 2^223 of the remainder sequence of (q, h) -- rho == tau * h
    (mod q), 0 <= rho < 2^223, 0 < |tau| < 2^223 -- for random and degenerate challenges."""
    import random
    import torch
    from _libs import Q
    rnd = random.Random(29)
    hs = [0, 1, 2, 3, Q - 1, Q - 2, 2**223, 2**223 - 1, 2**223 + 1, 2**224, (Q - 1) // 2, (Q + 1) // 2, 2**445, 2**300 + 1]
    hs += [rnd.getrandbits(b) for b in (10, 64, 100, 222, 223, 224, 225, 300, 440)]
    hs += [(Q // k) % Q for k in (3, 5, 7, 2**30 + 1, 2**31 - 1, 2**62 + 1, 2**100 + 7, 2**222 + 1)]   # huge first quotients
    hs += [rnd.getrandbits(446) % Q for _ in range(20000)]
    n = len(hs)
    h = torch.from_numpy(_gen.scalars_from_ints(hs).view(np.int64)).cuda()
    rho = torch.zeros((n, 15), dtype=torch.int32, device="cuda")
    tau = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    ga.dev("half_size_pair", rho.data_ptr(), tau.data_ptr(), h.data_ptr(), n, None)
    torch.cuda.synchronize()
    rw = rho.cpu().numpy().view(np.uint32)
    tw = tau.cpu().numpy().view(np.uint32)
    for i, hv in enumerate(hs):
        r = sum(int(rw[i, k]) << (32 * k) for k in range(15))
        t = sum(int(tw[i, k]) << (32 * k) for k in range(8))
        if t >> 255:
            t -= 1 << 256
        r0, r1, t0, t1 = Q, hv, 0, 1
        while r1 >= 2**223:
            k = r0 // r1
            r0, r1, t0, t1 = r1, r0 - k * r1, t1, t0 - k * t1
        assert (r, t) == (r1, t1), hex(hv)
        assert t != 0 and abs(t) < 2**223 and r < 2**223 and (r - t * hv) % Q == 0
