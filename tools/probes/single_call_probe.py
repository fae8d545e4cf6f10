"""Wall clock of the remaining single-operation entry points (host call, copies included).   python tools/probes/single_call_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libgoldilocks_amd as ga, _gen

k = _gen.stream_scalars(2, b"scp/k")
pts = ga.precomputed_scalarmul_batch(k)
enc = ga.point_encode_batch(pts[:1])
e57 = ga.point_encode_like_eddsa_batch(pts[:1])
h = np.frombuffer(_gen.stream(b"scp/h", 112), np.uint8).reshape(1, 112).copy()


def t(fn, label):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print("single call %-32s %.3f ms" % (label, (time.perf_counter() - t0) / 20 * 1e3), flush=True)


t(lambda: ga.point_encode_batch(pts[:1]), "point_encode")
t(lambda: ga.point_decode_batch(enc), "point_decode")
t(lambda: ga.point_encode_like_eddsa_batch(pts[:1]), "point_mul_by_ratio_and_encode_like_eddsa")
t(lambda: ga.point_decode_like_eddsa_batch(e57), "point_decode_like_eddsa_and_mul_by_ratio")
t(lambda: ga.point_from_hash_batch(h[:, :56]), "point_from_hash_nonuniform")
t(lambda: ga.point_from_hash_batch(h, uniform=True), "point_from_hash_uniform")
t(lambda: ga.precompute(pts[0]), "precompute")
t(lambda: ga.direct_scalarmul_batch(enc, k[:1]), "direct_scalarmul")
t(lambda: ga.point_dual_scalarmul_batch(pts[:1], k[:1], k[1:2]), "point_dual_scalarmul")
