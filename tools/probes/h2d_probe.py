"""Timing probe: host <-> device copies of 221 MiB (the input of 2^20 verifications) from pageable and pinned host memory,
and one core's memcpy: what bounds the host-array entry points is not the transfers (52 - 57 GB/s either way).
python tools/probes/h2d_probe.py"""
import torch, numpy as np, time
n = 221 * 1024 * 1024
a = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
t = torch.from_numpy(a)
for name, src in (("pageable", t), ("pinned", t.pin_memory())):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(src, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "H2D 221 MiB: %.2f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
h = torch.empty(n, dtype=torch.uint8)
for name, dst in (("pageable", h), ("pinned", torch.empty(n, dtype=torch.uint8).pin_memory())):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "D2H 221 MiB: %.2f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
import os
b = np.empty_like(a)
t0 = time.perf_counter(); np.copyto(b, a); dt = time.perf_counter() - t0
print("host memcpy 1 thread: %.1f GB/s; cpus %d" % (n / dt / 1e9, len(os.sched_getaffinity(0))))
