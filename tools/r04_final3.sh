#!/bin/bash
# the round's closing call, once more, on the final library (24-bit base table, 5 x 9 x 10 combs, three rounds of S*B
# ahead): the GPU suite, smoke, the profile round, the driver's own bench command, a soak
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.txt 2>&1; tail -4 gpurun_out/final/gputest.txt
python -c "
import sys; sys.path.insert(0, '.')
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -3
bash tools/profile_round.sh r04 > gpurun_out/final/profile.log 2>&1; tail -2 gpurun_out/final/profile.log
cp gpurun_out/profiles_r04/pmc_traffic.json profiles/pmc_traffic.json
bash tools/r04_call21.sh
bash tools/soak.sh ${1:-20} r04f
