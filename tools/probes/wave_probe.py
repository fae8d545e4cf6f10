"""Quick check + timing of the one-operation-per-wave kernel (tests/test_gpu_wave.py is the real test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga
d = np.load(os.path.join(ROOT, "tests", "golden", "f1_varbase.npz"))
bases, st = ga.point_decode_batch(d["base"], allow_identity=True)
for mx in (0, 1 << 20):
    ga.set_wave_batch_max(mx)
    got = ga.point_encode_batch(ga.point_scalarmul_batch(bases, d["scalar"]))
    bad = np.nonzero((got != d["out"]).any(axis=1))[0]
    print("wave_batch_max", mx, "mismatches", len(bad), bad[:10], flush=True)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
rng = np.random.default_rng(0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 1 << 16
idx = rng.integers(0, 1024, N)
B, S = dv(bases[idx]), dv(d["scalar"][idx]); out = torch.empty_like(B)
print("%8s %12s %12s" % ("n", "wave ms", "lane ms"))
for lg in (0, 2, 4, 6, 8, 10, 11, 12, 13, 14, 15, 16):
    n = 1 << lg
    r = []
    for mx in (1 << 20, 0):
        ga.set_wave_batch_max(mx)
        r.append(timeit(lambda: ga.dev("point_scalarmul", out.data_ptr(), B.data_ptr(), S.data_ptr(), n, None)))
    print("%8d %12.3f %12.3f" % (n, r[0], r[1]), flush=True)

# verification: one per wave vs one per lane
from _libs import oracle
import _gen
O = oracle()
sigs, pks, msgs = _gen.signatures(O, 4096, msglen=32, seed=b"wprobe-sig", nkeys=64)
sigs[5, 9] ^= 1
dsig, dpk = torch.from_numpy(sigs).cuda(), torch.from_numpy(pks).cuda()
dmsg = torch.from_numpy(np.frombuffer(b"".join(msgs), np.uint8).reshape(4096, 32).copy()).cuda()
st = torch.empty(4096, dtype=torch.int32, device="cuda")
print("%8s %12s %12s   verify" % ("n", "wave ms", "lane ms"))
for lg in (0, 4, 8, 10, 11, 12):
    n = 1 << lg
    r = []
    for mx in (1 << 20, 0):
        ga.set_wave_batch_max(mx)
        r.append(timeit(lambda: ga.dev("ed448_verify", st.data_ptr(), dsig.data_ptr(), dpk.data_ptr(), dmsg.data_ptr(), None, 32, 0, None, 0, n, None)))
        s_ = st[:n].cpu().numpy()
        assert (s_[np.arange(n) != 5] == -1).all() and (n <= 5 or s_[5] == 0), (mx, n, s_[:8])
    print("%8d %12.3f %12.3f" % (n, r[0], r[1]), flush=True)

# single drop-in calls, wall clock (host launch + sync included)
import time, ctypes as C
L_ = ga.lib()
pt = bases[:1].copy(); sc1 = d["scalar"][:1].copy(); outp = np.zeros((1, 32), np.uint64)
for mx, name in ((8192, "wave"), (0, "lane")):
    ga.set_wave_batch_max(mx)
    for fn, label in ((lambda: L_.goldilocks_448_point_scalarmul(outp.ctypes.data, pt.ctypes.data, sc1.ctypes.data), "point_scalarmul"),
                      (lambda: L_.goldilocks_448_base_double_scalarmul_non_secret(outp.ctypes.data, sc1.ctypes.data, pt.ctypes.data, sc1.ctypes.data), "base_double_scalarmul"),
                      (lambda: ga.ed448_verify(sigs[0].tobytes(), pks[0].tobytes(), msgs[0]), "ed448_verify")):
        fn(); t0 = time.perf_counter()
        for _ in range(20): fn()
        print("single call %-24s %-5s %.3f ms" % (label, name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
ga.set_wave_batch_max(8192)

# the other single calls
sk1 = np.frombuffer(_gen.stream(b"wprobe/sk", 57), np.uint8).reshape(1, 57).copy()
xs1 = np.frombuffer(_gen.stream(b"wprobe/x", 56), np.uint8).reshape(1, 56).copy()
pk1 = ga.ed448_derive_public_key_batch(sk1)
pub1, _ = ga.x448_batch(xs1)
k1 = d["scalar"][:1].copy()
for mx, name in ((8192, "wave"), (0, "lane")):
    ga.set_wave_batch_max(mx)
    for fn, label in ((lambda: ga.precomputed_scalarmul_batch(k1), "precomputed_scalarmul"),
                      (lambda: ga.ed448_derive_public_key_batch(sk1), "ed448_derive_public_key"),
                      (lambda: ga.ed448_sign_batch(sk1, pk1, [b"hello"]), "ed448_sign"),
                      (lambda: ga.x448_batch(xs1), "x448_derive_public_key"),
                      (lambda: ga.x448_batch(xs1, pub1), "x448"),
                      (lambda: ga.point_dual_scalarmul_batch(pt, sc1, sc1), "point_dual_scalarmul")):   # direct_scalarmul: tests/direct_probe.py
        fn(); t0 = time.perf_counter()
        for _ in range(20): fn()
        print("single call %-24s %-5s %.3f ms" % (label, name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
ga.set_wave_batch_max(8192)
