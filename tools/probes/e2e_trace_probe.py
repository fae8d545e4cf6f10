"""Probe (with a library built -DGD_TRACE_E2E=1, GOLDILOCKS_AMD_LIB=...): the host-array verification call by call, the
library printing its own laps.  python tools/probes/e2e_trace_probe.py"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make
n = 1 << 20
sig, pk, msg = make(n, 1024)
sig_h, pk_h, msg_h = sig.cpu().numpy(), pk.cpu().numpy(), msg.cpu().numpy()
L = ga.lib()
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
mptr = (msg_h.ctypes.data + 32 * np.arange(n, dtype=np.uint64)).astype(np.uint64)
mlen = np.full(n, 32, dtype=np.uint64)
st = np.zeros(n, dtype=np.int32)
for rep in range(4):
    t0 = time.perf_counter()
    rc = L.goldilocks_ed448_verify_batch(ptr(st), ptr(sig_h), ptr(pk_h), ptr(mptr), ptr(mlen), 0, None, 0, n)
    print("call %d: rc %d, %.2f ms, accepted %d" % (rep, rc, (time.perf_counter() - t0) * 1e3, int((st == -1).sum())), flush=True)
