"""The DRIVER's command (bench.py with no flags: headline + configs) over library variants on one box, interleaved:
    python tools/probes/default_run_ab.py [passes] [variant names...]
Prints the headline and every config's kernel time per library (GOLDILOCKS_AMD_LIB; an older library lacks newer entry points,
which the binding tolerates for a variant)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
names = sys.argv[2:]
libs = {"product": os.path.join(ROOT, "libgoldilocks_amd", "libgoldilocks_amd.so")}
for f in sorted(glob.glob(os.path.join(ROOT, "variants", "libgoldilocks_amd_*.so"))):
    libs[os.path.basename(f)[len("libgoldilocks_amd_"):-3]] = f
for p in range(passes):
    for name, lib in libs.items():
        if names and name not in names:
            continue
        env = dict(os.environ, GOLDILOCKS_AMD_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-end-to-end"], env=env, capture_output=True, text=True)
        try:
            line = json.loads(r.stdout.strip().splitlines()[-1])
            print("pass %d %-10s headline %.3f ms  " % (p, name, line["ms_per_step"]) +
                  "  ".join("%s %.3f" % (k, c["kernel_ms_avg"]) for k, c in line["configs"].items()), flush=True)
        except Exception:   # noqa
            print(name, "FAILED", r.stdout[-300:], r.stderr[-500:], flush=True)
