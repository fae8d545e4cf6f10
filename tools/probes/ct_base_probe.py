"""Timing probe: the base-point operations with index-independent table access (the library default) at 2^20:
base-point multiplication, key derivation, signing, X448 key generation.   python tools/probes/ct_base_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

N = 1 << 20
ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
scal = d(_gen.stream_scalars(N, b"ctp/scalar"))
out = torch.empty((N, 32), dtype=torch.int64, device="cuda")
sk = torch.from_numpy(np.frombuffer(_gen.stream(b"ctp/sk", 57 * N), np.uint8).reshape(N, 57).copy()).cuda()
pk = torch.empty((N, 57), dtype=torch.uint8, device="cuda")
msg = torch.from_numpy(np.frombuffer(_gen.stream(b"ctp/msg", 32 * N), np.uint8).reshape(N, 32).copy()).cuda()
sig = torch.empty((N, 114), dtype=torch.uint8, device="cuda")
xs = torch.from_numpy(np.frombuffer(_gen.stream(b"ctp/x", 56 * N), np.uint8).reshape(N, 56).copy()).cuda()
xo = torch.empty((N, 56), dtype=torch.uint8, device="cuda")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (("precomputed_scalarmul(base)", lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), None, scal.data_ptr(), N, None)),
                 ("ed448_derive_public_key", lambda: ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), N, None)),
                 ("ed448_sign", lambda: ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, N, None)),
                 ("x448_derive_public_key", lambda: ga.dev("x448", xo.data_ptr(), None, None, xs.data_ptr(), N, None))):
    t = timeit(fn)
    print("%-30s %8.3f ms  %8.1f M/s" % (name, t, N / t / 1e3), flush=True)
