"""Timing probe: 2^20 signatures of 2^10 ... 2^16 keys with library defaults (combs): what the entries kernel's segment
size costs where the keys are few (latency) and many (throughput).  python tools/probes/comb_seg_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make, timeit
n = 1 << 20
for nk in (1 << 10, 1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16):
    sig, pk, msg = make(n, nk)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    t = timeit(f); assert int((st == -1).sum()) == n
    print("n=2^20 keys=2^%-2d  %7.3f ms   %s" % (nk.bit_length() - 1, t, ga.last_verify_key_counts(teeth=True)), flush=True)
