"""Manual probe: the base point's window table at every width -- time to build it (the first call that needs it), its
size, and the base point's multiplication through it.   python tools/probes/base_table_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import libgoldilocks_amd as ga
import _gen
assert ga.lib().goldilocks_amd_init(0) == 0
s = _gen.stream_scalars(1 << 16, b"btp")
ref = None
print("%5s %10s %12s %14s" % ("bits", "MiB", "build ms", "2^16 mults ms"))
for bits in (16, 8, 12, 18, 20, 22, 24, 0):
    ga.set_base_table_bits(bits)
    free0, _ = torch.cuda.mem_get_info()
    t0 = time.perf_counter()
    out = ga.precomputed_scalarmul_batch(s[:64], flags=ga.CALL_TABLES_FAST)
    t1 = time.perf_counter()
    free1, _ = torch.cuda.mem_get_info()
    t2 = time.perf_counter()
    out = ga.point_encode_batch(ga.precomputed_scalarmul_batch(s, flags=ga.CALL_TABLES_FAST))
    t3 = time.perf_counter()
    if ref is None:
        ref = out
    got = ga.get_base_table_bits()
    entries = -(-446 // got) << (got - 1)
    print("%5s %10.1f %12.1f %14.2f  %s" % ("%d" % got if bits else "0->%d" % got, entries * 192 / 2**20, (t1 - t0) * 1e3, (t3 - t2) * 1e3,
                                         "same bytes" if (out == ref).all() else "MISMATCH"), flush=True)
