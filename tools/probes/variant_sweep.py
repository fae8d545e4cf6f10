"""Time bench workloads over experimental library builds (variants/*.so, tools/build_variants.py) on the GPU box:
    python tools/probes/variant_sweep.py "<bench args>" [variant names...]      (no names: every variant + the product)
One bench.py subprocess per (variant); prints value and kernel_ms_avg."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1].split()
names = sys.argv[2:]
libs = {"product": os.path.join(ROOT, "libgoldilocks_amd", "libgoldilocks_amd.so")}
for f in sorted(glob.glob(os.path.join(ROOT, "variants", "libgoldilocks_amd_*.so"))):
    libs[os.path.basename(f)[len("libgoldilocks_amd_"):-3]] = f
for name, lib in libs.items():
    if names and name not in names:
        continue
    env = dict(os.environ, GOLDILOCKS_AMD_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-configs", "--no-end-to-end"] + args,
                       env=env, capture_output=True, text=True)
    try:
        line = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-28s %-40s %12.0f /s  kernel %.3f ms  %s" % (name, " ".join(args), line["value"], line["roofline"]["kernel_ms_avg"],
                                                           line["config"]["parity_spot_check"]), flush=True)
    except Exception as e:   # noqa
        print(name, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)
