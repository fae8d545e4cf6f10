"""Timing probe: pooled window tables (combs off) with the batch's signatures in random key order and sorted by key.
python tools/probes/pooled_order_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import timeit
def make(n, nk, sort):
    sk = np.frombuffer(_gen.stream(b"pp/sk", 57 * nk), np.uint8).reshape(nk, 57)
    pk_k = ga.ed448_derive_public_key_batch(sk)
    key_of = np.random.default_rng(5).integers(0, nk, n)
    if sort: key_of = np.sort(key_of)
    msg = np.frombuffer(_gen.stream(b"pp/msg", 32 * n), np.uint8).reshape(n, 32).copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_sk, d_pk, d_msg = d(sk[key_of]), d(pk_k[key_of]), d(msg)
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_sign", sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    return sig, d_pk, d_msg
ga.set_verify_key_combs(0, 1)
n = 1 << 20
for nk in (1 << 12, 1 << 15, 1 << 17):
    t = []
    for sort in (False, True):
        sig, pk, msg = make(n, nk, sort)
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        t.append(timeit(f)); assert int((st == -1).sum()) == n and ga.last_verify_key_counts()[1] > 0
    print("n=2^20 keys=%-7d pooled tables: random order %7.3f ms   sorted by key %7.3f ms   %+.1f %%" % (nk, t[0], t[1], 100 * (t[1] - t[0]) / t[0]), flush=True)
ga.set_verify_key_combs()
