"""Config 4 (2^20 signatures of 2^10 keys) under the front kernel's experiment knobs, interleaved passes on one box:
    python tools/probes/ahead_probe.py [passes] [lib ...]
GOLDILOCKS_AMD_AHEAD_ROUNDS = 0 (the whole first range, small blocks: the product) or k rounds on one persistent block
per CU; GOLDILOCKS_AMD_AHEAD_HASH = 1 / 0 (challenges ahead as well, or S*B alone)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
libs = sys.argv[2:] or [os.path.join(ROOT, "libgoldilocks_amd", "libgoldilocks_amd.so")]
configs = [(0, 1), (0, 0), (3, 1), (4, 1), (5, 1), (6, 1), (4, 0)]
if os.environ.get("AHEAD_PROBE_CONFIGS"):     # "0:1,4:1"
    configs = [tuple(int(x) for x in c.split(":")) for c in os.environ["AHEAD_PROBE_CONFIGS"].split(",")]
best = {}
for p in range(passes):
    for lib in libs:
        for rounds, hashed in configs:
            env = dict(os.environ, GOLDILOCKS_AMD_LIB=lib, GOLDILOCKS_AMD_AHEAD_ROUNDS=str(rounds), GOLDILOCKS_AMD_AHEAD_HASH=str(hashed))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "verify", "--steps", "30", "--no-cpu-baseline",
                                "--no-configs", "--no-end-to-end"], env=env, capture_output=True, text=True)
            try:
                line = json.loads(r.stdout.strip().splitlines()[-1])
                ms = line["roofline"]["kernel_ms_avg"]
            except Exception:   # noqa
                print("FAILED", r.stdout[-200:], r.stderr[-400:], flush=True)
                continue
            key = (os.path.basename(lib), rounds, hashed)
            best.setdefault(key, []).append(ms)
            print("pass %d  %-34s rounds %d hash %d   %.3f ms" % (p, key[0], rounds, hashed, ms), flush=True)
print()
for key, v in best.items():
    print("%-34s rounds %d hash %d   min %.3f  median %.3f  (%s)" % (key[0], key[1], key[2], min(v), sorted(v)[len(v) // 2],
                                                                  " ".join("%.3f" % x for x in v)))
