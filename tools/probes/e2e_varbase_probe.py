"""Manual probe: goldilocks_448_point_scalarmul_batch from host arrays, 2^20 operations, seven calls (profiles/r04/experiments.md N).
   python tools/probes/e2e_varbase_probe.py"""
import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
L = ga.lib()
n = 1 << 20
k = _gen.stream_scalars(n, b"e2evb/k")
s = _gen.stream_scalars(n, b"e2evb/s")
bases = ga.precomputed_scalarmul_batch(k)          # n points
out = np.empty_like(bases)
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
ts = []
for rep in range(7):
    t0 = time.perf_counter()
    rc = L.goldilocks_448_point_scalarmul_batch(ptr(out), ptr(bases), ptr(s), n)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
print(os.environ.get("GOLDILOCKS_AMD_VARBASE_LATER_CHUNKS", "product"), "schedule: median %.2f ms  %s" % (sorted(ts)[3], " ".join("%.2f" % t for t in ts)), flush=True)
