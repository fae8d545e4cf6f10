"""Timing probe: verification with every lane for itself, with the pool of per-key window tables, and with per-key combs,
batch sizes 2^16 .. 2^20, 16 keys .. one key per signature (sampled with replacement).  python tools/probes/key_pool_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make, timeit
QUICK = "--quick" in sys.argv     # 2^20 signatures around the combs' break-even only
SMALL = "--small" in sys.argv     # ... and 2^16 .. 2^18 signatures around theirs
for n in ((1 << 20,) if QUICK else (1 << 16, 1 << 17, 1 << 18) if SMALL else (1 << 16, 1 << 17, 1 << 18, 1 << 20)):
    for nk in ((1024, n // 32, n // 16, n // 8) if QUICK else (n // 64, n // 32, n // 16, n // 8) if SMALL else (16, 1024, n // 128, n // 32, n // 16, n // 8, n // 4, n)):
        sig, pk, msg = make(n, nk)
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        ga.set_verify_key_combs(0, 1)
        ga.set_verify_key_pool(0, 0); a = timeit(f)
        ga.set_verify_key_pool(ga.KEY_POOL_DEFAULT, 0); b = timeit(f); assert int((st == -1).sum()) == n
        ga.set_verify_key_combs(1 << 17, 1); c = timeit(f) if nk <= 1 << 17 and nk * 4 <= n else float("nan"); assert int((st == -1).sum()) == n
        ga.set_verify_key_combs(); d = timeit(f); assert int((st == -1).sum()) == n
        print("n=2^%d keys=%-7d (%6.1f per key)  every lane for itself %7.3f ms   pooled tables %7.3f ms   combs forced %7.3f ms   library default %7.3f ms"
              % (n.bit_length() - 1, nk, n / nk, a, b, c, d), flush=True)
ga.set_verify_key_pool()
ga.set_verify_key_combs()
