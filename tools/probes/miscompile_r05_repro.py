"""One command to ask a toolchain whether it still miscompiles round 5's block (docs/history/r05.md H):

    python tools/probes/miscompile_r05_repro.py            (on a box with a gfx950 device; builds the variant if missing)

Builds the library with -DGD_REPRO_R05_DIVERGENT_INVERSION=1 -- goldilocks_448_direct_scalarmul's fallback for an encoding
that does not decode computes u(B) with a field inversion INSIDE the divergent block again, as it did until commit
2976ced -- and runs every lane of a 300-operation batch with three undecodable encodings against the oracle, in place and
rolled by one wave.  Prints the toolchain, the kernel's register figures and REPRODUCED (the lanes that differ) or NOT
REPRODUCED.  The product never contains this block (the macro is off; u(B) is a generated constant)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANT = os.path.join(ROOT, "variants", "libgoldilocks_amd_repro_r05.so")
if not os.path.exists(VARIANT):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_variants.py"), "repro_r05:-DGD_REPRO_R05_DIVERGENT_INVERSION=1"])
os.environ["GOLDILOCKS_AMD_LIB"] = VARIANT
import numpy as np                      # noqa: E402
import libgoldilocks_amd as ga          # noqa: E402
import _gen                             # noqa: E402
from _libs import Scalar, oracle        # noqa: E402

O = oracle()
ga.set_wave_batch_max(0)                # the lane kernel, not one operation per wave
print("toolchain:", subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0:2])
res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), VARIANT], capture_output=True, text=True).stdout
print([l for l in res.splitlines() if l.startswith("k_direct_scalarmul_ct") or l.startswith("kernel")])
n = 300
s = _gen.random_scalars(n, b"t-direct-s")
base = _gen.oracle_encode(_gen.oracle_fixed(O, _gen.random_scalars(n, b"t-direct-b")))
base[5] = 0
base[6] = 0xff
base[7, 0] |= 1                         # three encodings that do not decode: the base point is multiplied instead
bad = []
for roll in (0, 64):
    perm = np.roll(np.arange(n), roll)
    got, st = ga.direct_scalarmul_batch(base[perm], s[perm], allow_identity=False, short_circuit=False)
    for k, i in enumerate(perm):
        out = (C.c_uint8 * 56)()
        r = O.orc_direct_scalarmul(out, base[i].ctypes.data, C.cast(s[i].ctypes.data, C.POINTER(Scalar)), 0, 0)
        if r != st[k] or bytes(out) != got[k].tobytes():
            bad.append((roll, k, int(i)))
print("REPRODUCED: (roll, position, input) of the lanes that differ from the oracle: %s" % bad if bad else
      "NOT REPRODUCED: every lane equals the oracle's (this toolchain compiles the divergent block correctly)")
