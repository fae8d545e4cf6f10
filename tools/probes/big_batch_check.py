"""Manual check (not collected by pytest): 2^23+12345 operations through the device API, and 3 000 000\nthrough the sharded, pipelined host API, against the oracle on a sample.  python tools/probes/big_batch_check.py"""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from _libs import oracle
O = oracle()
n = (1 << 23) + 12345
k = _gen.stream_scalars(n, b"big/base"); s = _gen.stream_scalars(n, b"big/scalar")
d = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
dk, ds = d(k), d(s)
bases = torch.empty((n, 32), dtype=torch.int64, device="cuda"); out = torch.empty_like(bases)
ga.dev("precomputed_scalarmul", bases.data_ptr(), None, dk.data_ptr(), n, None)
ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), ds.data_ptr(), n, None)
st = torch.empty(n, dtype=torch.int32, device="cuda")
ga.dev("point_pred", st.data_ptr(), out.data_ptr(), None, 1, n, None)
assert int((st == -1).sum()) == n
idx = np.concatenate([np.random.default_rng(1).integers(0, n, 500), [0, n - 1, n - 12345, 1 << 23]])
got = out.cpu().numpy().view(np.uint64)[idx]; b_h = bases.cpu().numpy().view(np.uint64)[idx]
assert (ga.point_encode_batch(got) == _gen.oracle_encode(_gen.oracle_varbase(O, b_h, s[idx]))).all()
# host API with sharding over "3 devices" and pipelining at this size
ga.use_devices([0, 0, 0])
h = ga.point_scalarmul_batch(bases.cpu().numpy().view(np.uint64)[:3000000], s[:3000000])
ga.use_devices(None)
assert (h == out.cpu().numpy().view(np.uint64)[:3000000]).all()
print("big batch ok", n)
