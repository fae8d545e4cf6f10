"""Timing probe: goldilocks_amd_direct_scalarmul_dev for small batches (one operation per wave).   python tools/probes/direct_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

N = 4096
k = torch.from_numpy(_gen.stream_scalars(N, b"dp/k").view(np.int64)).cuda()
s = torch.from_numpy(_gen.stream_scalars(N, b"dp/s").view(np.int64)).cuda()
pts = torch.empty((N, 32), dtype=torch.int64, device="cuda")
ga.dev("precomputed_scalarmul", pts.data_ptr(), None, k.data_ptr(), N, None)
enc = torch.empty((N, 56), dtype=torch.uint8, device="cuda")
ga.dev("point_encode", enc.data_ptr(), pts.data_ptr(), N, None)
out = torch.empty((N, 56), dtype=torch.uint8, device="cuda")
st = torch.empty(N, dtype=torch.int32, device="cuda")
for mx in (8192, 0):
    ga.set_wave_batch_max(mx)
    for n in (1, 64, 1024, 4096):
        fn = lambda: ga.dev("direct_scalarmul", out.data_ptr(), st.data_ptr(), enc.data_ptr(), s.data_ptr(), 0, 0, n, None)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("wave_batch_max %5d  n %5d  %.3f ms" % (mx, n, e0.elapsed_time(e1) / 5), flush=True)
ga.set_wave_batch_max(8192)
