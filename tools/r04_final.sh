#!/bin/bash
# the round's closing call on the final library: the GPU suite, the profile round, an extended soak
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/final/gputest.txt 2>&1; tail -4 gpurun_out/final/gputest.txt
bash tools/profile_round.sh ${1:-r04} > gpurun_out/final/profile.log 2>&1; tail -3 gpurun_out/final/profile.log
bash tools/soak.sh ${2:-40} ${3:-r04b}
