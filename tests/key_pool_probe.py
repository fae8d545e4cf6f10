"""Timing probe: verification with every lane for itself, with the pool of per-key window tables, and with per-key combs,
batch sizes 2^16 .. 2^20, 16 keys .. one key per signature (sampled with replacement).  python tests/key_pool_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
def make(n, nk):
    sk = np.frombuffer(_gen.stream(b"pp/sk", 57 * nk), np.uint8).reshape(nk, 57)
    pk_k = ga.ed448_derive_public_key_batch(sk)
    key_of = np.random.default_rng(5).integers(0, nk, n)
    msg = np.frombuffer(_gen.stream(b"pp/msg", 32 * n), np.uint8).reshape(n, 32).copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_sk, d_pk, d_msg = d(sk[key_of]), d(pk_k[key_of]), d(msg)
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_sign", sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    return sig, d_pk, d_msg
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1 << 16, 1 << 17, 1 << 18, 1 << 20):
    for nk in (16, 1024, n // 128, n // 32, n // 16, n // 8, n // 4, n):
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
