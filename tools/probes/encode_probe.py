"""Timing probe: the encode / decode kernels at 2^20 points (device-resident).   python tools/probes/encode_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

N = 1 << 20
k = torch.from_numpy(_gen.stream_scalars(N, b"ep/k").view(np.int64)).cuda()
pts = torch.empty((N, 32), dtype=torch.int64, device="cuda")
ga.dev("precomputed_scalarmul", pts.data_ptr(), None, k.data_ptr(), N, None)
e56 = torch.empty((N, 56), dtype=torch.uint8, device="cuda")
e57 = torch.empty((N, 57), dtype=torch.uint8, device="cuda")
out = torch.empty_like(pts)
st = torch.empty(N, dtype=torch.int32, device="cuda")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (("point_encode (decaf)", lambda: ga.dev("point_encode", e56.data_ptr(), pts.data_ptr(), N, None)),
                 ("point_decode (decaf)", lambda: ga.dev("point_decode", out.data_ptr(), st.data_ptr(), e56.data_ptr(), 0, N, None)),
                 ("encode_like_eddsa, shared inversions", lambda: ga.dev("point_encode_eddsa", e57.data_ptr(), pts.data_ptr(), N, None)),
                 ("encode_like_eddsa, 8 launches of 2^17 (one inversion per point)",
                  lambda: [ga.dev("point_encode_eddsa", e57.data_ptr() + 57 * (i << 17), pts.data_ptr() + 256 * (i << 17), 1 << 17, None) for i in range(8)]),
                 ("decode_like_eddsa", lambda: ga.dev("point_decode_eddsa", out.data_ptr(), st.data_ptr(), e57.data_ptr(), N, None))):
    t = timeit(fn)
    print("%-66s %8.3f ms  %8.1f M/s" % (name, t, N / t / 1e3), flush=True)
