"""Manual probe: does a verification step that follows an idle device run slower than one that follows another step?
(The host-array pipeline's first passes run their rounds a fifth slower than the resident launch's: profiles/r04/experiments.md
G reads that as clocks that came down while the keys were uploaded.)   python tools/probes/idle_gap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make
n = 1 << 20
sig, pk, msg = make(n, 1024)
st = torch.empty(n, dtype=torch.int32, device="cuda")
f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
for _ in range(5):
    f()
torch.cuda.synchronize()
print("%10s %12s" % ("idle ms", "step ms"))
for gap in (0, 0.5, 1, 2, 5, 20, 100, 0):
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        if gap:
            time.sleep(gap * 1e-3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%10.1f %12.3f   (min %.3f, max %.3f)" % (gap, sorted(ts)[len(ts) // 2], min(ts), max(ts)), flush=True)
assert int((st == -1).sum()) == n
