#!/usr/bin/env python3
"""Generate the golden fixtures from the REAL reference (oracle/_ref, arch_ref64 build).

Runs only in the build container (needs /root/reference compiled by `make -C oracle ref`).
Everything written here is DATA: seeded inputs and the reference's outputs.

  f1_varbase.npz    1024 x {base (56-B decaf encoding), scalar (56 B), scalar*base (56-B encoding)}
                    incl. edge scalars and identity / base-point bases       (BASELINE config 1)
  f2_fixed.npz      512 x {scalar, scalar*B encoding}; 64 x the same on a random precomputed point
  f3_verify.npz     256 Ed448 verify cases (valid + corrupted + malformed) with the reference's verdict
  f4_field.npz      512 x {a, b, a*b, a^2, isr(a), isr mask} as canonical 56-byte strings, with
                    unreduced-limb inputs
  f5_constants.json base point limbs, SHA-256 of the comb table, sizeof/alignof
  f6_bench_digest.json  SHAKE256 digests over the 2^k outputs of the benchmark input stream
  f7_verify_torsion.json  48 verify cases whose R / public key carry 2- or 4-torsion components or are
                    small-order points themselves, with the reference's verdict
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _gen
from _libs import ref, Point, Scalar, Q, P, ORACLE_DIR

R = ref()
F = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_field.so"))
F.ref_gf_isr.restype = C.c_uint64
vp = lambda a: a.ctypes.data_as(C.c_void_p)


def ref_from_hash(raw112):
    p = Point()
    R.goldilocks_448_point_from_hash_uniform(C.byref(p), (C.c_uint8 * 112).from_buffer_copy(raw112))
    return p


def ref_encode(p):
    b = (C.c_uint8 * 56)()
    R.goldilocks_448_point_encode(b, C.byref(p))
    return bytes(b)


def ref_scalar(v):
    return Scalar.from_int(v % Q)


def f1():
    n = 1024
    raw = _gen.stream(b"golden/f1/points", 112 * n)
    sc = _gen.random_scalars(n, b"golden/f1/scalars")
    edge = [0, 1, 2, Q - 1, Q - 2, 2**445, 2**445 - 1, (Q + 1) // 2, 31, 32, 2**224, 2**440 + 12345]
    sc[:len(edge)] = _gen.scalars_from_ints(edge)
    base_enc, out_enc = np.empty((n, 56), np.uint8), np.empty((n, 56), np.uint8)
    for i in range(n):
        p = ref_from_hash(raw[112 * i:112 * i + 112])
        if i == 20:
            p = Point.from_buffer_copy(bytes(Point.in_dll(R, "goldilocks_448_point_identity")))
        if i == 21:
            p = Point.from_buffer_copy(bytes(R.point_base))
        s = Scalar.from_buffer_copy(sc[i].tobytes())
        o = Point()
        R.goldilocks_448_point_scalarmul(C.byref(o), C.byref(p), C.byref(s))
        base_enc[i] = np.frombuffer(ref_encode(p), np.uint8)
        out_enc[i] = np.frombuffer(ref_encode(o), np.uint8)
    np.savez_compressed(os.path.join(HERE, "f1_varbase.npz"), base=base_enc, scalar=sc, out=out_enc)


def f2():
    n = 512
    sc = _gen.random_scalars(n, b"golden/f2/scalars")
    sc[:6] = _gen.scalars_from_ints([0, 1, Q - 1, 2**445, 2, 2**446 - 1])
    out = np.empty((n, 56), np.uint8)
    for i in range(n):
        o = Point()
        R.goldilocks_448_precomputed_scalarmul(C.byref(o), R.precomputed_base, C.byref(Scalar.from_buffer_copy(sc[i].tobytes())))
        out[i] = np.frombuffer(ref_encode(o), np.uint8)
    pt = ref_from_hash(_gen.stream(b"golden/f2/point", 112))
    tab = (C.c_uint8 * 15360)()
    tab_al = C.addressof(tab)
    R.goldilocks_448_precompute(tab, C.byref(pt))
    sc2 = _gen.random_scalars(64, b"golden/f2/scalars2")
    out2 = np.empty((64, 56), np.uint8)
    for i in range(64):
        o = Point()
        R.goldilocks_448_precomputed_scalarmul(C.byref(o), tab, C.byref(Scalar.from_buffer_copy(sc2[i].tobytes())))
        out2[i] = np.frombuffer(ref_encode(o), np.uint8)
    np.savez_compressed(os.path.join(HERE, "f2_fixed.npz"), scalar=sc, out=out,
                        point=np.frombuffer(ref_encode(pt), np.uint8),
                        table=np.frombuffer(bytes(tab), np.uint64), scalar2=sc2, out2=out2)


def f3():
    n = 256
    rng = np.random.default_rng(3)
    sigs, pks, msgs, ctxs, phs, verdict, kinds = [], [], [], [], [], [], []
    for i in range(n):
        sk = _gen.stream(b"golden/f3/sk%d" % (i % 32), 57)
        pk = (C.c_uint8 * 57)()
        R.goldilocks_ed448_derive_public_key(pk, (C.c_uint8 * 57).from_buffer_copy(sk))
        mlen = [0, 1, 11, 12, 32, 125, 126, 135, 136, 137, 300][i % 11]
        msg = _gen.stream(b"golden/f3/msg%d" % i, max(mlen, 1))[:mlen]
        ctx = _gen.stream(b"golden/f3/ctx%d" % i, 255)[:[0, 0, 3, 255][i % 4]]
        ph = 1 if i % 7 == 3 else 0
        mb = (C.c_uint8 * max(1, mlen)).from_buffer_copy(msg or b"\0")
        cb = (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0")
        sig = (C.c_uint8 * 114)()
        R.goldilocks_ed448_sign(sig, (C.c_uint8 * 57).from_buffer_copy(sk), pk, mb, mlen, ph, cb, len(ctx))
        sig, pk, msg = bytearray(sig), bytearray(pk), bytearray(msg)
        kind = ["valid", "flip_R", "flip_S", "flip_pk", "flip_msg", "bad_R_byte56", "bad_pk_byte56", "R_noncanonical",
                "S_plus_q", "pk_noncanonical", "flip_ctx", "valid", "R_all_ff", "valid", "S_high_bits", "valid"][i % 16]
        if kind == "flip_R": sig[rng.integers(0, 57)] ^= 1 << rng.integers(0, 8)
        if kind == "flip_S": sig[57 + rng.integers(0, 56)] ^= 1 << rng.integers(0, 8)
        if kind == "flip_pk": pk[rng.integers(0, 56)] ^= 1 << rng.integers(0, 8)
        if kind == "flip_msg" and mlen: msg[rng.integers(0, mlen)] ^= 1
        if kind == "bad_R_byte56": sig[56] |= 0x01 << rng.integers(0, 7)
        if kind == "bad_pk_byte56": pk[56] |= 0x01 << rng.integers(0, 7)
        if kind == "R_noncanonical":   # y + p does not fit 448 bits except for tiny y: use y = p + small => >= p
            sig[0:56] = (P + int(rng.integers(0, 1000))).to_bytes(56, "little")
        if kind == "pk_noncanonical": pk[0:56] = (P + int(rng.integers(0, 1000))).to_bytes(56, "little")
        if kind == "S_plus_q": sig[57:114] = (int.from_bytes(sig[57:114], "little") + Q).to_bytes(57, "little")
        if kind == "S_high_bits": sig[113] |= 0x80
        if kind == "R_all_ff": sig[0:57] = b"\xff" * 57
        if kind == "flip_ctx" and len(ctx): ctx = bytes([ctx[0] ^ 1]) + ctx[1:]
        cb = (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0")
        mb = (C.c_uint8 * max(1, mlen)).from_buffer_copy(bytes(msg) or b"\0")
        v = R.goldilocks_ed448_verify((C.c_uint8 * 114).from_buffer_copy(sig), (C.c_uint8 * 57).from_buffer_copy(pk),
                                      mb, mlen, ph, cb, len(ctx))
        sigs.append(bytes(sig)); pks.append(bytes(pk)); msgs.append(bytes(msg)); ctxs.append(ctx); phs.append(ph)
        verdict.append(v); kinds.append(kind)
    json.dump({"cases": [{"sig": s.hex(), "pk": p.hex(), "msg": m.hex(), "ctx": c.hex(), "prehashed": ph, "verdict": v,
                          "kind": k} for s, p, m, c, ph, v, k in zip(sigs, pks, msgs, ctxs, phs, verdict, kinds)]},
              open(os.path.join(HERE, "f3_verify.json"), "w"), indent=0)
    print("f3 verdicts:", {k: sorted(set(v for v, kk in zip(verdict, kinds) if kk == k)) for k in set(kinds)})


def f4():
    n = 512
    rng = np.random.default_rng(4)
    a = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    b = rng.integers(0, 2**56, size=(n, 8), dtype=np.uint64)
    a[0] = 0; b[0] = 0
    a[1] = 2**56 - 1; b[1] = 2**56 - 1
    a[2] = 2**56 + 255; b[2] = 2**56 + 255            # weakly reduced, limbs above 2^56
    a[3] = [1, 0, 0, 0, 0, 0, 0, 0]
    a[4] = [2**56 - 1] * 4 + [2**56 - 2] + [2**56 - 1] * 3   # p
    a[5] = [2**56 - 2] + [2**56 - 1] * 3 + [2**56 - 2] + [2**56 - 1] * 3   # p - 1
    res = {k: np.empty((n, 56), np.uint8) for k in ("a", "b", "mul", "sqr", "isr")}
    mask = np.empty(n, np.uint8)
    o = (C.c_uint64 * 8)()
    ser = (C.c_uint8 * 56)()
    for i in range(n):
        def put(key, limbs):
            F.ref_gf_serialize(ser, limbs)
            res[key][i] = np.frombuffer(bytes(ser), np.uint8)
        put("a", vp(a[i])); put("b", vp(b[i]))
        F.ref_gf_mul(o, vp(a[i]), vp(b[i])); put("mul", o)
        F.ref_gf_sqr(o, vp(a[i])); put("sqr", o)
        mask[i] = 1 if F.ref_gf_isr(o, vp(a[i])) else 0
        put("isr", o)
    np.savez_compressed(os.path.join(HERE, "f4_field.npz"), a_limbs=a, b_limbs=b, isr_mask=mask, **res)


def f5():
    pre = C.string_at(R.precomputed_base.value, 15360)
    base = bytes(R.point_base)
    json.dump({"point_base_limbs": [int(x) for x in np.frombuffer(base, np.uint64)],
               "point_base_encoding": ref_encode(Point.from_buffer_copy(base)).hex(),
               "precomputed_base_sha256": hashlib.sha256(pre).hexdigest(),
               "sizeof_precomputed_s": int(C.c_size_t.in_dll(R, "goldilocks_448_sizeof_precomputed_s").value),
               "alignof_precomputed_s": int(C.c_size_t.in_dll(R, "goldilocks_448_alignof_precomputed_s").value),
               "sizeof_point_s": C.sizeof(Point), "sizeof_scalar_s": C.sizeof(Scalar),
               "scalar_q": hex(Q)},
              open(os.path.join(HERE, "f5_constants.json"), "w"), indent=1)


def f6():
    """Digest of the reference's outputs on the benchmark input stream (rank 0): bases = k*B,
    out = s*base; digest = SHAKE256 over the concatenated 56-byte encodings of the first 2^k outputs."""
    from _libs import oracle
    O = oracle()   # oracle == reference bit for bit (tests/test_oracle_vs_ref.py); used here for its thread pool
    n = 1 << 20
    k = _gen.stream_scalars(n, b"bench_varbase_v1/0/base")
    s = _gen.stream_scalars(n, b"bench_varbase_v1/0/scalar")
    bases = _gen.oracle_fixed(O, k)
    out = np.empty((n, 32), np.uint64)
    fn = C.cast(R.goldilocks_448_point_scalarmul, C.c_void_p)     # the real reference does the work
    O.orc_extern_scalarmul_batch(fn, vp(out), vp(bases), vp(s), n, _gen.NTHREADS)
    enc = _gen.oracle_encode(out)
    # spot-check the oracle encoder against the reference encoder
    for i in range(0, n, 65537):
        assert ref_encode(Point.from_buffer_copy(out[i].tobytes())) == enc[i].tobytes()
    dig = {str(lg): hashlib.shake_256(enc[:1 << lg].tobytes()).hexdigest(32) for lg in (10, 14, 16, 18, 20)}
    json.dump({"label": "bench_varbase_v1/0", "digest_shake256_32": dig,
               "how": "SHAKE256(concat of 56-byte decaf encodings of out[0:2^k]), k in keys"},
              open(os.path.join(HERE, "f6_bench_digest.json"), "w"), indent=1)


# ---------------------------------------------------------------------------------------------------
# f7: small-order (torsion) malleability.  Signatures whose R (or whose public key) carries a
# 2- or 4-torsion component satisfy the verification equation only up to that component; whether they
# are accepted is a property of the reference's point comparison (X1*Y2 == Y1*X2 on the isogenous
# curve), so the reference's verdict is captured as data.  The signatures are made with independent
# Python big-integer Ed448 arithmetic (RFC 8032 section 5.2), not with any library under test.
ED_D = -39081
ED_BX = 224580040295924300187604334099896036246789641632564134246125461686950415467406032909029192869357953282578032075146446173674602635247710
ED_BY = 298819210078481492676017930443930673437544040154080242095928241372331506189835876003536878655418784733982303233503462500531545062832660


def ed_add(P1, P2):
    (x1, y1), (x2, y2) = P1, P2
    t = ED_D * x1 * x2 * y1 * y2 % P
    return ((x1 * y2 + y1 * x2) * pow(1 + t, P - 2, P) % P, (y1 * y2 - x1 * x2) * pow(1 - t, P - 2, P) % P)


def ed_mul(k, Pt):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = ed_add(acc, Pt)
        Pt = ed_add(Pt, Pt)
        k >>= 1
    return acc


def ed_enc(Pt):
    x, y = Pt
    return y.to_bytes(56, "little") + bytes([0x80 if x & 1 else 0])


def shake114(*parts):
    return hashlib.shake_256(b"".join(parts)).digest(114)


def f7():
    assert (ED_BX * ED_BX + ED_BY * ED_BY - 1 - ED_D * ED_BX * ED_BX * ED_BY * ED_BY) % P == 0
    B = (ED_BX, ED_BY)
    torsion = {"none": (0, 1), "T2": (0, P - 1), "T4": (1, 0), "T4neg": (P - 1, 0)}
    cases = []
    for i in range(40):
        sk = _gen.stream(b"golden/f7/sk%d" % (i % 8), 57)
        msg = _gen.stream(b"golden/f7/msg%d" % i, 40)[:[0, 7, 40][i % 3]]
        ctx = b"" if i % 2 else b"f7"
        h = shake114(sk)
        a_bytes = bytearray(h[:57]); a_bytes[0] &= 0xfc; a_bytes[55] |= 0x80; a_bytes[56] = 0
        a = int.from_bytes(a_bytes, "little")
        A = ed_mul(a, B)
        pk_ref = (C.c_uint8 * 57)()
        R.goldilocks_ed448_derive_public_key(pk_ref, (C.c_uint8 * 57).from_buffer_copy(sk))
        assert bytes(pk_ref) == ed_enc(A), "python Ed448 disagrees with the reference's derive_public_key"
        kind_R, kind_A = [("none", "none"), ("T2", "none"), ("T4", "none"), ("T4neg", "none"), ("none", "T2"),
                          ("none", "T4"), ("T2", "T2"), ("T4", "T4neg")][i % 8]
        dom = b"SigEd448" + bytes([0, len(ctx)]) + ctx
        r = int.from_bytes(shake114(dom, h[57:], msg), "little") % Q
        A_pub = ed_add(A, torsion[kind_A])                # the key the verifier is given
        R_pub = ed_add(ed_mul(r, B), torsion[kind_R])     # the R the verifier is given
        hh = int.from_bytes(shake114(dom, ed_enc(R_pub), ed_enc(A_pub), msg), "little") % Q
        S = (r + hh * a) % Q
        sig = ed_enc(R_pub) + S.to_bytes(57, "little")
        if kind_R == "none" and kind_A == "none":         # sanity: this is exactly RFC 8032 signing
            ref_sig = (C.c_uint8 * 114)()
            R.goldilocks_ed448_sign(ref_sig, (C.c_uint8 * 57).from_buffer_copy(sk), pk_ref,
                                    (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg or b"\0"), len(msg), 0,
                                    (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0"), len(ctx))
            assert bytes(ref_sig) == sig, "python Ed448 signing disagrees with the reference"
        v = R.goldilocks_ed448_verify((C.c_uint8 * 114).from_buffer_copy(sig), (C.c_uint8 * 57).from_buffer_copy(ed_enc(A_pub)),
                                      (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg or b"\0"), len(msg), 0,
                                      (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0"), len(ctx))
        cases.append({"sig": sig.hex(), "pk": ed_enc(A_pub).hex(), "msg": msg.hex(), "ctx": ctx.hex(), "prehashed": 0,
                      "verdict": v, "kind": "R+%s,A+%s" % (kind_R, kind_A)})
    # degenerate encodings as keys / R: identity, order 2, order 4
    base_case = cases[0]
    for name, pt in torsion.items():
        for where in ("R", "pk"):
            sig, pk = bytearray.fromhex(base_case["sig"]), bytearray.fromhex(base_case["pk"])
            if where == "R":
                sig[:57] = ed_enc(pt)
            else:
                pk[:] = ed_enc(pt)
            msg, ctx = bytes.fromhex(base_case["msg"]), bytes.fromhex(base_case["ctx"])
            v = R.goldilocks_ed448_verify((C.c_uint8 * 114).from_buffer_copy(sig), (C.c_uint8 * 57).from_buffer_copy(pk),
                                          (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg or b"\0"), len(msg), 0,
                                          (C.c_uint8 * max(1, len(ctx))).from_buffer_copy(ctx or b"\0"), len(ctx))
            cases.append({"sig": bytes(sig).hex(), "pk": bytes(pk).hex(), "msg": msg.hex(), "ctx": ctx.hex(), "prehashed": 0,
                          "verdict": v, "kind": "%s=%s" % (where, name)})
    json.dump({"cases": cases}, open(os.path.join(HERE, "f7_verify_torsion.json"), "w"), indent=0)
    kinds = sorted(set(c["kind"] for c in cases))
    print("f7 verdicts:", {k: sorted(set(c["verdict"] for c in cases if c["kind"] == k)) for k in kinds})


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7"]
    for w in which:
        print("generating", w, flush=True)
        globals()[w]()
