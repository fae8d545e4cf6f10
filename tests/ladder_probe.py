"""Timing probe: one-table against two-table ladders at 2^20 (fast tables), for profiles/<round>/experiments.md.
python tests/ladder_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
N = 1 << 20
s1 = d(_gen.stream_scalars(N, b"probe/s1")); s2 = d(_gen.stream_scalars(N, b"probe/s2"))
b1 = torch.empty((N, 32), dtype=torch.int64, device="cuda"); b2 = torch.empty_like(b1); out = torch.empty_like(b1)
ga.dev("precomputed_scalarmul", b1.data_ptr(), None, s1.data_ptr(), N, None)
ga.dev("precomputed_scalarmul", b2.data_ptr(), None, s2.data_ptr(), N, None)
ga.set_table_access(ga.TABLES_FAST)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = timeit(lambda: ga.dev("point_scalarmul", out.data_ptr(), b1.data_ptr(), s1.data_ptr(), N, None))
b = timeit(lambda: ga.dev("point_double_scalarmul", out.data_ptr(), b1.data_ptr(), s1.data_ptr(), b2.data_ptr(), s2.data_ptr(), N, None))
print("point_scalarmul         %.3f ms   (660 632 MACs)" % a)
print("point_double_scalarmul  %.3f ms   (660 632 + 26 K table + 90 x 1 536 = 825 K MACs): %.4f of the MAC-proportional time" % (b, b / (a * 825.0 / 660.6)))
ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
