"""Diagnostic: goldilocks_448_direct_scalarmul batch outputs of the loaded library (GOLDILOCKS_AMD_LIB) against the
oracle, every lane, both identity rules; prints the lanes that differ.  python tools/probes/direct_diff_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libgoldilocks_amd as ga, _gen
from _libs import oracle, Scalar
O = oracle()
ga.set_wave_batch_max(0)
n = 300
s = _gen.random_scalars(n, b"t-direct-s")
pts = _gen.oracle_fixed(O, _gen.random_scalars(n, b"t-direct-b"))
base = _gen.oracle_encode(pts)
base[5] = 0; base[6] = 0xff; base[7, 0] |= 1
for allow_id in (False, True):
    got, st = ga.direct_scalarmul_batch(base, s, allow_identity=allow_id, short_circuit=False)
    bad = []
    for i in range(n):
        out = (C.c_uint8 * 56)()
        r = O.orc_direct_scalarmul(out, base[i].ctypes.data, C.cast(s[i].ctypes.data, C.POINTER(Scalar)), 1 if allow_id else 0, 0)
        if r != st[i] or bytes(out) != got[i].tobytes():
            bad.append((i, r, int(st[i])))
    print("allow_identity", allow_id, "mismatching lanes:", bad)
# the same operations again with the failing inputs moved to other lanes
perm = np.roll(np.arange(n), 64)
got, st = ga.direct_scalarmul_batch(base[perm], s[perm], allow_identity=False, short_circuit=False)
bad = []
for k, i in enumerate(perm):
    out = (C.c_uint8 * 56)()
    r = O.orc_direct_scalarmul(out, base[i].ctypes.data, C.cast(s[i].ctypes.data, C.POINTER(Scalar)), 0, 0)
    if r != st[k] or bytes(out) != got[k].tobytes():
        bad.append((k, int(i)))
print("rolled by 64: mismatching (position, original index):", bad)
