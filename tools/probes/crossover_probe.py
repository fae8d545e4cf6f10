"""Timing probe: one-operation-per-wave against lane kernels around the dispatch thresholds (device-resident).
python tools/probes/crossover_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

N = 16384
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
k = d(_gen.stream_scalars(N, b"xo/k")); s = d(_gen.stream_scalars(N, b"xo/s"))
pts = torch.empty((N, 32), dtype=torch.int64, device="cuda"); out = torch.empty_like(pts); out2 = torch.empty_like(pts)
ga.dev("precomputed_scalarmul", pts.data_ptr(), None, k.data_ptr(), N, None)
sk = torch.from_numpy(np.frombuffer(_gen.stream(b"xo/sk", 57 * N), np.uint8).reshape(N, 57).copy()).cuda()
pk = torch.empty((N, 57), dtype=torch.uint8, device="cuda")
msg = torch.from_numpy(np.frombuffer(_gen.stream(b"xo/msg", 32 * N), np.uint8).reshape(N, 32).copy()).cuda()
sig = torch.empty((N, 114), dtype=torch.uint8, device="cuda")
st = torch.empty(N, dtype=torch.int32, device="cuda")
xs = torch.from_numpy(np.frombuffer(_gen.stream(b"xo/x", 56 * N), np.uint8).reshape(N, 56).copy()).cuda()
xo = torch.empty((N, 56), dtype=torch.uint8, device="cuda"); xp = torch.empty((N, 56), dtype=torch.uint8, device="cuda")
e56 = torch.empty((N, 56), dtype=torch.uint8, device="cuda")
tabs = torch.empty((4096, 1920), dtype=torch.int64, device="cuda")
ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), N, None)
ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, N, None)
ga.dev("x448", xp.data_ptr(), None, None, xs.data_ptr(), N, None)
ga.dev("point_encode", e56.data_ptr(), pts.data_ptr(), N, None)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4


ops = {
    "point_scalarmul": lambda n: ga.dev("point_scalarmul", out.data_ptr(), pts.data_ptr(), s.data_ptr(), n, None),
    "precomputed_scalarmul(base)": lambda n: ga.dev("precomputed_scalarmul", out.data_ptr(), None, s.data_ptr(), n, None),
    "point_double_scalarmul": lambda n: ga.dev("point_double_scalarmul", out.data_ptr(), pts.data_ptr(), s.data_ptr(), pts.data_ptr(), k.data_ptr(), n, None),
    "point_dual_scalarmul": lambda n: ga.dev("point_dual_scalarmul", out.data_ptr(), out2.data_ptr(), pts.data_ptr(), s.data_ptr(), k.data_ptr(), n, None),
    "direct_scalarmul": lambda n: ga.dev("direct_scalarmul", xo.data_ptr(), st.data_ptr(), e56.data_ptr(), s.data_ptr(), 0, 0, n, None),
    "ed448_verify": lambda n: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None),
    "ed448_sign": lambda n: ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None),
    "ed448_derive_public_key": lambda n: ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), n, None),
    "x448": lambda n: ga.dev("x448", xo.data_ptr(), st.data_ptr(), xp.data_ptr(), xs.data_ptr(), n, None),
    "x448_derive_public_key": lambda n: ga.dev("x448", xo.data_ptr(), None, None, xs.data_ptr(), n, None),
    "precompute": lambda n: ga.dev("precompute", tabs.data_ptr(), pts.data_ptr(), min(n, 4096), None),
}
print("%-30s %s" % ("ms: wave / lane at n =", "      1024            2048            4096            8192"))
for name, fn in ops.items():
    row = []
    for n in (1024, 2048, 4096, 8192):
        ga.set_wave_batch_max(1 << 20); a = timeit(lambda: fn(n))
        ga.set_wave_batch_max(0); b = timeit(lambda: fn(n))
        row.append("%6.3f / %6.3f" % (a, b))
    print("%-30s %s" % (name, "  ".join(row)), flush=True)
ga.set_wave_batch_max(8192)
