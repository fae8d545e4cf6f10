#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call6
mkdir -p "$OUT"
cd "$ROOT"
GOLDILOCKS_AMD_TRACE=1 timeout 200 python tests/e2e_trace_probe.py > "$OUT/e2e_laps.txt" 2>&1
tail -45 "$OUT/e2e_laps.txt"
