"""Timing probe: verification through per-key combs of 7, 8 and 9 teeth (4 x 7 x 16, 4 x 8 x 14, 5 x 9 x 10), 2^18 .. 2^21
signatures of 2^8 .. 2^13 keys.
python tools/probes/wide_comb_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make, timeit
for n in (1 << 18, 1 << 20, 1 << 21):
    for nk in (256, 1024, 2048, 4096, 8192):
        sig, pk, msg = make(n, nk)
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        ga.set_verify_key_combs_xwide(0)
        ga.set_verify_key_combs_wide(0); a = timeit(f); assert int((st == -1).sum()) == n and ga.last_verify_key_counts(teeth=True)[3] == 7
        ga.set_verify_key_combs_wide(1); b = timeit(f); assert int((st == -1).sum()) == n and ga.last_verify_key_counts(teeth=True)[3] == 8
        ga.set_verify_key_combs_xwide(1); x = timeit(f); assert int((st == -1).sum()) == n and ga.last_verify_key_counts(teeth=True)[3] == 9
        print("n=2^%d keys=%-5d (%6.1f per key)  7 teeth %7.3f ms   8 teeth %7.3f ms  %+.1f %%   5 x 9 teeth %7.3f ms  %+.1f %% against 8"
              % (n.bit_length() - 1, nk, n / nk, a, b, 100 * (b - a) / a, x, 100 * (x - b) / b), flush=True)
ga.set_verify_key_combs_wide()
ga.set_verify_key_combs_xwide()
