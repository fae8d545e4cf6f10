#!/bin/bash
# the round's closing call on the final library (24-bit base table by default): the GPU suite, smoke, an extended soak --
# half of its input streams with the default table, half with a 20-bit table (another span of the recoding)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.txt 2>&1; tail -6 gpurun_out/final/gputest.txt
python -c "
import sys; sys.path.insert(0, '.')
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -3
bash tools/soak.sh ${1:-20} r04c
GOLDILOCKS_AMD_BASE_TABLE_BITS=20 bash tools/soak.sh ${1:-20} r04d
