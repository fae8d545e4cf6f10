"""Manual check: how long does the first call (context + table build) take?  python tools/probes/init_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import libgoldilocks_amd as ga
L = ga.lib()
t0 = time.perf_counter(); assert L.goldilocks_amd_init(0) == 0; t1 = time.perf_counter()
print("goldilocks_amd_init: %.1f ms" % ((t1 - t0) * 1e3))
L.goldilocks_amd_shutdown()
t0 = time.perf_counter(); assert L.goldilocks_amd_init(0) == 0; t1 = time.perf_counter()
print("again after shutdown: %.1f ms" % ((t1 - t0) * 1e3), ga.device_info())
