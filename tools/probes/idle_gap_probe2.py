"""Manual probe, second part: what keeps the clocks up across a gap?  The gap before a verification step is filled with
(a) nothing, (b) a stream of tiny kernels (a 4-KiB fill, back to back), (c) one block per CU of dependent arithmetic
(a small matmul chain).   python tools/probes/idle_gap_probe2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make
n = 1 << 20
sig, pk, msg = make(n, 1024)
st = torch.empty(n, dtype=torch.int32, device="cuda")
f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
tiny = torch.zeros(1024, device="cuda")
a = torch.randn(512, 512, device="cuda"); b = torch.randn(512, 512, device="cuda")
big = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
def fill_none(ms):
    time.sleep(ms * 1e-3)
def fill_tiny(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        tiny.add_(1.0)
def fill_small_mm(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        torch.mm(a, b)
def fill_big_mm(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        torch.mm(big, big)
        torch.cuda.synchronize()
for _ in range(5):
    f()
torch.cuda.synchronize()
print("%-14s %8s %10s" % ("gap filled by", "gap ms", "step ms"))
for name, fill in (("nothing", fill_none), ("tiny kernels", fill_tiny), ("512^2 matmuls", fill_small_mm), ("8192^2 bf16 mm", fill_big_mm)):
    for gap in (1, 5, 20):
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            fill(gap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-14s %8.1f %10.3f   (min %.3f, max %.3f)" % (name, gap, sorted(ts)[len(ts) // 2], min(ts), max(ts)), flush=True)
