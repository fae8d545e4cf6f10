"""Manual probe: the first verification steps of a process, one by one (the base table is built inside the first; how
long until the step time settles?).   python tools/probes/step_ramp_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make
n = 1 << 20
sig, pk, msg = make(n, 1024)
st = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
ts = []
for i in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    ts.append((e0, e1))
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ts]
print("steps 1..40 (ms):", " ".join("%.2f" % x for x in ms))
print("bits", ga.get_base_table_bits())
