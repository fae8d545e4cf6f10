"""Timing probe: the variable-base entry points at 2^20 in both table-access modes (per-call flags), for
profiles/<round>/.  python tools/probes/ct_varbase_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
N = 1 << 20
s1 = d(_gen.stream_scalars(N, b"probe/s1")); s2 = d(_gen.stream_scalars(N, b"probe/s2"))
b1 = torch.empty((N, 32), dtype=torch.int64, device="cuda"); b2 = torch.empty_like(b1)
o1 = torch.empty_like(b1); o2 = torch.empty_like(b1)
ga.dev("precomputed_scalarmul", b1.data_ptr(), None, s1.data_ptr(), N, None)
ga.dev("precomputed_scalarmul", b2.data_ptr(), None, s2.data_ptr(), N, None)
enc = torch.empty((N, 56), dtype=torch.uint8, device="cuda"); eo = torch.empty_like(enc)
st = torch.empty(N, dtype=torch.int32, device="cuda")
ga.dev("point_encode", enc.data_ptr(), b1.data_ptr(), N, None)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, flags in (("index-independent (default)", ga.CALL_TABLES_INDEX_INDEPENDENT), ("fast", ga.CALL_TABLES_FAST)):
    print(name)
    t = timeit(lambda: ga.dev("point_scalarmul", o1.data_ptr(), b1.data_ptr(), s1.data_ptr(), N, None, flags=flags))
    print("  point_scalarmul          %7.3f ms  %6.2f M/s" % (t, N / t / 1e3))
    t = timeit(lambda: ga.dev("direct_scalarmul", eo.data_ptr(), st.data_ptr(), enc.data_ptr(), s1.data_ptr(), 0, 0, N, None, flags=flags))
    print("  direct_scalarmul         %7.3f ms  %6.2f M/s" % (t, N / t / 1e3))
    t = timeit(lambda: ga.dev("point_dual_scalarmul", o1.data_ptr(), o2.data_ptr(), b1.data_ptr(), s1.data_ptr(), s2.data_ptr(), N, None, flags=flags), reps=3)
    print("  point_dual_scalarmul     %7.3f ms  %6.2f M/s" % (t, N / t / 1e3))
    t = timeit(lambda: ga.dev("point_double_scalarmul", o1.data_ptr(), b1.data_ptr(), s1.data_ptr(), b2.data_ptr(), s2.data_ptr(), N, None, flags=flags), reps=3)
    print("  point_double_scalarmul   %7.3f ms  %6.2f M/s" % (t, N / t / 1e3))
