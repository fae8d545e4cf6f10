"""Throughput and latency of the three headline kernels against the batch size (device-resident data),
for profiles/<round>/batch_sweep.txt.  python tools/probes/batch_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libgoldilocks_amd as ga, _gen

print(ga.device_info(), flush=True)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
N = 1 << 22
scal = d(_gen.stream_scalars(N, b"sweep/scalar"))
k = d(_gen.stream_scalars(N, b"sweep/base"))
bases = torch.empty((N, 32), dtype=torch.int64, device="cuda")
ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), N, None)
out = torch.empty_like(bases)
sk = torch.from_numpy(np.frombuffer(_gen.stream(b"sweep/sk", 57 * N), np.uint8).reshape(N, 57).copy()).cuda()
pk = torch.empty((N, 57), dtype=torch.uint8, device="cuda")
msg = torch.from_numpy(np.frombuffer(_gen.stream(b"sweep/msg", 32 * N), np.uint8).reshape(N, 32).copy()).cuda()
sig = torch.empty((N, 114), dtype=torch.uint8, device="cuda")
ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), N, None)
ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, N, None)
st = torch.empty(N, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("variable base: batches up to %d operations run one operation per wavefront (csrc/wave_coop.hpp)" % ga.get_wave_batch_max())
for mode, name in ((ga.TABLES_INDEX_INDEPENDENT, "index-independent tables (library default)"), (ga.TABLES_FAST, "fast tables (opt-in)"),
                   (ga.TABLES_INDEX_INDEPENDENT, "index-independent tables, one-operation-per-wave path disabled")):
  ga.set_table_access(mode)
  ga.set_wave_batch_max(0 if "disabled" in name else 8192)
  print("\n" + name)
  print("%8s | %21s | %21s | %21s" % ("batch", "variable-base", "base point", "verify (always fast)"))
  print("%8s | %10s %10s | %10s %10s | %10s %10s" % ("", "ms", "M op/s", "ms", "M op/s", "ms", "M op/s"))
  for lg in range(6, 23, 2):
      n = 1 << lg
      reps = 3 if lg >= 18 else 5
      a = timeit(lambda: ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), scal.data_ptr(), n, None), reps)
      b = timeit(lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), None, scal.data_ptr(), n, None), reps)
      c = timeit(lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0,
                                None, 0, n, None), reps)
      assert int((st[:n] == -1).sum()) == n
      print("    2^%-2d | %10.3f %10.3f | %10.3f %10.3f | %10.3f %10.3f" % (lg, a, n / a / 1e3, b, n / b / 1e3, c, n / c / 1e3),
            flush=True)

ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
ga.set_wave_batch_max(8192)
