"""First-contact probe for the GPU box: time the three batch kernels device-resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # a test-side script: it checks against the oracle
import numpy as np, torch
import libgoldilocks_amd as ga, _gen
from _libs import oracle
O = oracle()
print(ga.device_info(), flush=True)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
k = 1024
sk = _gen.random_scalars(k, b"probe-s"); bk = ga.precomputed_scalarmul_batch(_gen.random_scalars(k, b"probe-b"))
chk = ga.point_encode_batch(ga.point_scalarmul_batch(bk[:128], sk[:128]))
assert (chk == _gen.oracle_encode(_gen.oracle_varbase(O, bk[:128], sk[:128]))).all(); print("parity ok", flush=True)
rng = np.random.default_rng(0)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for logn in (14, 16, 17, 18, 20):
    n = 1 << logn
    bases = d(bk[rng.integers(0, k, n)]); scal = d(sk[rng.integers(0, k, n)]); out = torch.empty_like(bases)
    ms = timeit(lambda: ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), scal.data_ptr(), n, None))
    print("varbase  n=2^%d  %.2f ms  %.3f M/s" % (logn, ms, n / ms / 1e3), flush=True)
    ms = timeit(lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), None, scal.data_ptr(), n, None))
    print("fixed    n=2^%d  %.2f ms  %.3f M/s" % (logn, ms, n / ms / 1e3), flush=True)
sigs, pks, msgs = _gen.signatures(O, 4096, msglen=32, seed=b"probe-sig", nkeys=64)
for logn in (14, 17, 20):
    n = 1 << logn
    idx = rng.integers(0, 4096, n)
    ds, dp = torch.from_numpy(sigs[idx]).cuda(), torch.from_numpy(pks[idx]).cuda()
    dm = torch.from_numpy(np.frombuffer(b"".join(msgs), np.uint8).reshape(4096, 32)[idx].copy()).cuda()
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: ga.dev("ed448_verify", st.data_ptr(), ds.data_ptr(), dp.data_ptr(), dm.data_ptr(), None, 32, 0, None, 0, n, None))
    assert int((st == -1).sum()) == n
    print("verify   n=2^%d  %.2f ms  %.3f M/s" % (logn, ms, n / ms / 1e3), flush=True)

# PCIe-inclusive rate of the host-array API (pageable numpy buffers in, pre-touched buffer out)
import ctypes as C
n = 1 << 20
bh = np.ascontiguousarray(bk[rng.integers(0, k, n)]); sh = np.ascontiguousarray(sk[rng.integers(0, k, n)])
outh = np.zeros((n, 32), dtype=np.uint64)
L = ga.lib()
call = lambda: L.goldilocks_448_point_scalarmul_batch(outh.ctypes.data, bh.ctypes.data, sh.ctypes.data, n)
assert call() == 0
for rep in range(3):
    t0 = time.perf_counter(); assert call() == 0; dt = time.perf_counter() - t0
    print("varbase host-array API (H2D + kernel + D2H) n=2^20  %.1f ms  %.2f M/s" % (dt * 1e3, n / dt / 1e6), flush=True)
assert (ga.point_encode_batch(outh[:64]) == _gen.oracle_encode(_gen.oracle_varbase(O, bh[:64], sh[:64]))).all()

# PCIe-inclusive rate of the host-array verify API (message pointer tables packed on the host)
n = 1 << 20
idx = rng.integers(0, 4096, n)
sh_, ph_ = np.ascontiguousarray(sigs[idx]), np.ascontiguousarray(pks[idx])
mh_ = np.ascontiguousarray(np.frombuffer(b"".join(msgs), np.uint8).reshape(4096, 32)[idx])
ptrs = (C.c_void_p * n)(*[mh_.ctypes.data + 32 * i for i in range(n)])
lens = (C.c_size_t * n)(*([32] * n))
sth = np.zeros(n, np.int32)
call = lambda: L.goldilocks_ed448_verify_batch(sth.ctypes.data, sh_.ctypes.data, ph_.ctypes.data, ptrs, lens, 0, None, 0, n)
assert call() == 0 and (sth == -1).all()
for rep in range(3):
    t0 = time.perf_counter(); assert call() == 0; dt = time.perf_counter() - t0
    print("verify host-array API (pack + H2D + kernel + D2H) n=2^20  %.1f ms  %.2f M/s" % (dt * 1e3, n / dt / 1e6), flush=True)

# PCIe-inclusive rate of the host-array fixed-base API (56 B up, 256 B down per operation)
outf = np.zeros((n, 32), dtype=np.uint64)
call = lambda: L.goldilocks_448_precomputed_scalarmul_batch(outf.ctypes.data, C.c_void_p.in_dll(L, "goldilocks_448_precomputed_base"), sh.ctypes.data, n)
assert call() == 0
for rep in range(3):
    t0 = time.perf_counter(); assert call() == 0; dt = time.perf_counter() - t0
    print("fixed-base host-array API (H2D + kernel + D2H) n=2^20  %.1f ms  %.2f M/s" % (dt * 1e3, n / dt / 1e6), flush=True)
assert (ga.point_encode_batch(outf[:64]) == _gen.oracle_encode(_gen.oracle_fixed(O, sh[:64]))).all()
