"""Timing probe: goldilocks_448_base_double_scalarmul_non_secret at 2^20.   python tools/probes/base_double_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
N = 1 << 20
s1 = d(_gen.stream_scalars(N, b"probe/s1")); s2 = d(_gen.stream_scalars(N, b"probe/s2"))
b2 = torch.empty((N, 32), dtype=torch.int64, device="cuda"); out = torch.empty_like(b2)
ga.dev("precomputed_scalarmul", b2.data_ptr(), None, s2.data_ptr(), N, None)
fn = lambda: ga.dev("base_double_scalarmul", out.data_ptr(), s1.data_ptr(), b2.data_ptr(), s2.data_ptr(), N, None)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5
print("base_double_scalarmul_non_secret  %.3f ms  %.2f M/s" % (t, N / t / 1e3))
