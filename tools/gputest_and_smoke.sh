#!/bin/bash
# What the driver runs at round end, on the GPU box:   gpurun --timeout 2700 -- "bash tools/gputest_and_smoke.sh"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.txt 2>&1; tail -6 gpurun_out/final/gputest.txt
python -c "
import sys; sys.path.insert(0, '.')
import libgoldilocks_amd as ga
b = ga.build_info()
print('toolchain: ' + b['toolchain']); print('library_sha256: ' + b['library_sha256'])" >> gpurun_out/final/gputest.txt
python -c "
import sys; sys.path.insert(0, '.')
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -3
