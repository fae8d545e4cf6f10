"""Timing probe: verification of 2^13 .. 2^15 signatures of few keys with every lane for itself and with per-key combs
(the pool's minimum batch lowered for the occasion).  python tools/probes/small_batch_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make, timeit
for n in (1 << 13, 1 << 14, 1 << 15):
    for nk in (4, 64, n // 64, n // 16):
        sig, pk, msg = make(n, nk)
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        ga.set_verify_key_pool(0, 0); a = timeit(f); assert int((st == -1).sum()) == n
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT, 4097); ga.set_verify_key_combs(1 << 15, 1); b = timeit(f); assert int((st == -1).sum()) == n
        print("n=2^%d keys=%-5d (%6.1f per key)  every lane for itself %6.3f ms   combs %6.3f ms  %s" % (n.bit_length() - 1, nk, n / nk, a, b, ga.last_verify_key_counts()), flush=True)
ga.set_verify_key_pool(); ga.set_verify_key_combs()
