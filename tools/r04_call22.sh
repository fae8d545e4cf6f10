#!/bin/bash
# round-4 GPU call 22: a third geometry for the keys' combs, 5 x 9 x 10 (1 280 entries, 9 doublings + 49 additions per
# signature), for keys that sign a thousand signatures: parity, then the three geometries against each other
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call22
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_fullsize.py -x -q -k "verif or config5 or pipeline or keys or ten_thousand" 2>&1 | tail -5 | tee "$OUT/gputest.txt"
timeout 900 python tests/wide_comb_probe.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/wide_comb_probe.txt"
for x in 0 1024; do
  python - <<PY 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/bench_verify_xwide.txt"
import json, subprocess, sys, os
env = dict(os.environ)
code = "import libgoldilocks_amd as ga, runpy, sys; ga.set_verify_key_combs_xwide($x); sys.argv=['bench.py','--workload','verify','--steps','20','--warmup','5','--no-cpu-baseline','--no-configs','--no-end-to-end']; runpy.run_path('bench.py', run_name='__main__')"
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
l = json.loads(r.stdout.strip().splitlines()[-1])
print("xwide from", $x, "->", "%.1f M/s" % (l["value"] / 1e6), "kernel %.3f ms" % l["roofline"]["kernel_ms_avg"], l["roofline"]["kernel"], l["config"]["parity_spot_check"])
PY
done
